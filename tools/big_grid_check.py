#!/usr/bin/env python3
"""One-off sanity check far beyond Config 3 (index arithmetic, launch limits): 64 x 1024 x 8192
(4.3 GB per [z][m][k] tensor), two redshift rows against the CPU oracle."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import hmvec_amd as hm
from hmvec_amd.params import battaglia_defaults, default_params
from oracle import hmref
from conftest import power_close

NZ, NM, NK = (int(a) for a in (sys.argv[1:4] or (64, 1024, 8192)))
zs, ms, ks = np.linspace(0.01, 3.0, NZ), np.geomspace(2e10, 1e17, NM), np.geomspace(1e-4, 100, NK)
t0 = time.perf_counter()
h = hm.HaloModel(zs, ks, ms=ms, accuracy="low", engine="analytic")
h.add_battaglia_profile("electron", family="AGN", xmax=20, nxs=5000)
h.add_hod("g", mthresh=10 ** 10.5 + zs * 0.0)
pairs = [("nfw", "nfw"), ("electron", "electron"), ("g", "g"), ("nfw", "electron"), ("g", "nfw"), ("g", "electron")]
P = h.get_power_all(pairs)
print(f"GPU {NZ}x{NM}x{NK}: {time.perf_counter()-t0:.2f} s wall incl. first-use setup")
sel = np.array([1, NZ - 2])
z = zs[sel]
p = dict(default_params)
ksig = np.geomspace(p["sigma2_kmin"], p["sigma2_kmax"], p["sigma2_numks"])
ci = hmref.CosmoInputs(h=h.h, omm0=h.omm0, ombh2=p["ombh2"], rho_crit_0=float(h.rho_critical_z(0.0)),
                       rho_crit_zs=h.rho_critical_z(z), Pzk=h.Pzk[sel], sPzk=h.sPzk[sel], ks_sigma2=ksig,
                       h_of_z_zs=h.h_of_z(z))
o = hmref.RefHaloModel(ci, z, ks, ms, p)
o.add_battaglia_profile("electron", "AGN", p["battaglia_gas_gamma"], battaglia_defaults["AGN"], 5000, 20)
o.add_hod("g", mthresh=10 ** 10.5 + z * 0.0)
worst = 0.0
for a, b in pairs:
    ok, w = power_close(P[(a, b)][sel], o.get_power(a, b))
    assert ok, (a, b, w)
    worst = max(worst, w)
print("worst |dP|/tol", worst, " uk_e max abs err", np.max(np.abs(h.uk_profiles["electron"][sel] - o.uk_profiles["electron"])))
