"""Profile-stage time (HIP events) of the three long radial grids the reference's callers use, Config-3 grid:
gas (nxs=30000, xmax=50), numeric NFW (40000, 200), tSZ pressure (30000, 2) - for the library HMG_LIB_PATH names.
Also prints a checksum of each tensor (variants must agree to rounding).  Usage: python tools/probes/long_stage.py [label] [which]"""
import os, sys, numpy as np
sys.path.insert(0, ".")
import hmvec_amd as hm
from hmvec_amd import _native as nat
label = sys.argv[1] if len(sys.argv) > 1 else "main"
which = sys.argv[2] if len(sys.argv) > 2 else "gas,nfw,tsz"
zs = np.linspace(0.01, 3.0, 32); ms = np.geomspace(2e10, 1e17, 512); ks = np.geomspace(1e-4, 100, 4096)
ctx = nat.Context(0)
h = hm.HaloModel(zs, ks, ms=ms, accuracy="low", engine="analytic", ctx=ctx)
def stage(fn, name, reps=9):
    t = []
    for i in range(reps + 2):
        ctx.call("hmg_bracket_next", nat.KERNEL_PROFILE_FFT, 44, 45)
        fn()
        ctx.sync()
        if i >= 2:
            t.append(ctx.elapsed_ms(44, 45))
    return np.median(t), np.min(t)
out = [f"{label:10s}"]
if "gas" in which:
    m, lo = stage(lambda: h.add_battaglia_profile("electron", family="AGN", xmax=50, nxs=30000, ignore_existing=True), "gas")
    u = h.uk_profiles["electron"]; out.append(f"gas30000/50 {m:7.4f} (min {lo:6.4f}) sum {float(np.sum(u[::4, ::16])):.13e}")
if "nfw" in which:
    m, lo = stage(lambda: h.add_nfw_profile("nfwnum", numeric=True, ignore_existing=True), "nfw")
    u = h.uk_profiles["nfwnum"]; out.append(f"nfw40000/200 {m:7.4f} (min {lo:6.4f}) sum {float(np.sum(u[::4, ::16])):.13e}")
if "tsz" in which:
    m, lo = stage(lambda: h.add_battaglia_pres_profile("y", xmax=2, nxs=30000, ignore_existing=True), "tsz")
    u = h.pk_profiles["y"]; out.append(f"tsz30000/2 {m:7.4f} (min {lo:6.4f}) sum {float(np.sum(u[::4, ::16])):.13e}")
print("  ".join(out), flush=True)
ctx.close()
