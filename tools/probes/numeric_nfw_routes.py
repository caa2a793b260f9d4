"""add_nfw_profile(numeric=True) at the reference's defaults (nxs = 40000, xmax = 200; hmvec/params.py:59-60) on the
Config-3 grid: long-grid route against rocFFT.  HIP-event time of the profile stage, median of 5."""
import os, sys, numpy as np
sys.path.insert(0, ".")
import hmvec_amd as hm
from hmvec_amd import _native as nat
zs = np.linspace(0.01, 3.0, 32); ms = np.geomspace(2e10, 1e17, 512); ks = np.geomspace(1e-4, 100, 4096)
for label, env in (("long-grid route (LP=1250, R=16)", {}), ("long-grid route, no chirp (HMG_CHIRP=0)", {"HMG_CHIRP": "0"}), ("rocFFT route (HMG_PRUNED_FFT=0)", {"HMG_PRUNED_FFT": "0"})):
    os.environ.update(env)
    ctx = nat.Context(0)
    h = hm.HaloModel(zs, ks, ms=ms, accuracy="low", engine="analytic", ctx=ctx)
    t = []
    for i in range(7):
        ctx.call("hmg_bracket_next", nat.KERNEL_PROFILE_FFT, 44, 45)
        h.add_nfw_profile("nfwnum", numeric=True, ignore_existing=True)
        ctx.sync()
        if i >= 2:
            t.append(ctx.elapsed_ms(44, 45))
    u = h.uk_profiles["nfwnum"][::8, ::32]
    a = h.uk_profiles["nfw"][::8, ::32]
    print(f"numeric NFW nxs=40000 xmax=200: {label:45s} {np.median(t):8.3f} ms   max |u_numeric - u_analytic| on a sample {np.max(np.abs(u - a)):.2e}", flush=True)
    for k in env: os.environ.pop(k)
    ctx.close()
