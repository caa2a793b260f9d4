"""One long-grid profile stage on the Config-3 grid, a few times (for rocprofv3): gas | nfw | tsz."""
import sys, numpy as np
sys.path.insert(0, ".")
import hmvec_amd as hm
which = sys.argv[1]; reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
zs = np.linspace(0.01, 3.0, 32); ms = np.geomspace(2e10, 1e17, 512); ks = np.geomspace(1e-4, 100, 4096)
h = hm.HaloModel(zs, ks, ms=ms, accuracy="low", engine="analytic")
for _ in range(reps):
    if which == "gas":
        h.add_battaglia_profile("electron", family="AGN", xmax=50, nxs=30000, ignore_existing=True)
    elif which == "nfw":
        h.add_nfw_profile("nfwnum", numeric=True, ignore_existing=True)
    else:
        h.add_battaglia_pres_profile("y", family="pres", xmax=2, nxs=30000, ignore_existing=True)
    h._ctx().sync()
print("ok")
