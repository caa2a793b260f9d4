"""Only add_battaglia_pres_profile(xmax=2, nxs=30000) on the Config-3 grid, a few times (for rocprofv3)."""
import sys, numpy as np
sys.path.insert(0, ".")
import hmvec_amd as hm
zs = np.linspace(0.01, 3.0, 32); ms = np.geomspace(2e10, 1e17, 512); ks = np.geomspace(1e-4, 100, 4096)
h = hm.HaloModel(zs, ks, ms=ms, accuracy="low", engine="analytic")
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    h.add_battaglia_pres_profile("y", family="pres", xmax=2, nxs=30000, ignore_existing=True)
    h._ctx().sync()
print("ok", float(h.pk_profiles["y"][3, 100, 50]))
