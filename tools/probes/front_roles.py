"""Which role of the front launch sets its duration?  Runs passes with (a) the constructor only (sigma^2 | halo stage),
(b) + add_battaglia_profile (row parameters in the halo-stage threads), (c) + add_hod (occupation blocks) and prints
nothing: read the front_group_kernel averages from `rocprofv3 --kernel-trace --stats -- python3 tools/probes/front_roles.py a|b|c [nz]`."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import hmvec_amd as hm

variant = sys.argv[1] if len(sys.argv) > 1 else "c"
nz = int(sys.argv[2]) if len(sys.argv) > 2 else 32
zs, ms, ks = np.linspace(0.01, 3.0, nz), np.geomspace(2e10, 1e17, 512), np.geomspace(1e-4, 100, 4096)
h = hm.HaloModel(zs, ks, ms=ms, accuracy="low", engine="analytic")
thr = 10 ** 10.5 + zs * 0.0
for _ in range(60):
    h.init_mass_function(ms)
    h.add_nfw_profile("nfw", ignore_existing=True)
    if variant in "bc":
        h.add_battaglia_profile("electron", family="AGN", xmax=20, nxs=5000, ignore_existing=True)
    if variant == "c":
        h.add_hod("g", mthresh=thr, ignore_existing=True)
    h._flush() if hasattr(h, "_flush") else None
    h._ctx().sync()
