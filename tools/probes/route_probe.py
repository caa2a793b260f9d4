"""Which route does hmg_profile_fft take?  Times the profile stage of nxs = 10000 with and without HMG_FUSED_MAX_M."""
import os, sys, time
import numpy as np
sys.path.insert(0, ".")
import hmvec_amd as hm
from hmvec_amd import _native as nat
zs = np.array([0.1, 2.2]); ms = np.geomspace(2e10, 1e17, 40); ks = np.geomspace(1e-4, 100, 150)
res = []
for cap in (None, "2500"):
    if cap:
        os.environ["HMG_FUSED_MAX_M"] = cap
    ctx = nat.Context(0)
    h = hm.HaloModel(zs, ks, ms=ms, accuracy="low", engine="analytic", ctx=ctx)
    h.add_battaglia_profile("electron", family="AGN", xmax=20, nxs=10000)
    res.append(h.uk_profiles["electron"].copy())
    print(cap, "getenv", os.environ.get("HMG_FUSED_MAX_M"), res[-1][0, 20, 100:103])
print("equal", np.array_equal(res[0], res[1]), np.max(np.abs(res[0] - res[1])))
