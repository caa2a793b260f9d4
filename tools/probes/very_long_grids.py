import os, sys, numpy as np
sys.path.insert(0, ".")
import hmvec_amd as hm
from hmvec_amd import _native as nat
zs = np.array([0.1, 1.5]); ms = np.geomspace(2e10, 1e17, 12); ks = np.geomspace(1e-4, 100, 300)
out = {}
for nxs, xmax in ((200000, 400.0), (120000, 200.0), (60000, 4.0)):
    for route, env in (("long", {}), ("rocfft", {"HMG_PRUNED_FFT": "0", "HMG_BAND_FFT": "0"})):
        os.environ.update(env)
        ctx = nat.Context(0)
        h = hm.HaloModel(zs, ks, ms=ms, accuracy="low", engine="analytic", ctx=ctx)
        if xmax > 10:
            h.add_battaglia_profile("e", family="AGN", xmax=xmax, nxs=nxs)
            out[route] = h.uk_profiles["e"].copy()
        else:
            h.add_battaglia_pres_profile("y", xmax=xmax, nxs=nxs)
            out[route] = h.pk_profiles["y"].copy()
        for k in env: os.environ.pop(k)
        ctx.close()
    sc = np.max(np.abs(out["rocfft"]), axis=-1, keepdims=True)
    print(nxs, xmax, "max |d|/rowmax between routes:", float(np.max(np.abs(out["long"] - out["rocfft"]) / sc)), "finite:", bool(np.all(np.isfinite(out["long"]))))
