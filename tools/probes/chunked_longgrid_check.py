"""More rows than one 8-GiB block of scratch lines holds (the long-grid route then runs as several launches): numeric
NFW at the reference's defaults (nxs = 40000, xmax = 200: 160 KB of scratch line per row) on 64 x 1024 rows = 10.5 GB.
Rows on both sides of the launch boundary (row 53687) and the last rows against the oracle."""
import sys
import numpy as np
sys.path.insert(0, ".")
import hmvec_amd as hm
from hmvec_amd.params import default_params
from oracle import hmref

zs = np.linspace(0.05, 3.0, 64); ms = np.geomspace(2e10, 1e17, 1024); ks = np.geomspace(1e-3, 50, 128)
h = hm.HaloModel(zs, ks, ms=ms, accuracy="low", engine="analytic", skip_nfw=True)
_, u = h.add_nfw_profile("nfwnum", numeric=True)
u = np.asarray(u)
print("tensor", u.shape, "finite", bool(np.all(np.isfinite(u))))
p = dict(default_params)
ksig = np.geomspace(p["sigma2_kmin"], p["sigma2_kmax"], p["sigma2_numks"])
worst = 0.0
for zsel, msel in ((slice(52, 54), slice(424, 440)), (slice(62, 64), slice(1016, 1024)), (slice(0, 2), slice(0, 8))):
    z, m = zs[zsel], ms[msel]
    ci = hmref.CosmoInputs(h=h.h, omm0=h.omm0, ombh2=p["ombh2"], rho_crit_0=float(h.rho_critical_z(0.0)),
                           rho_crit_zs=h.rho_critical_z(z), Pzk=h.Pzk[zsel], sPzk=h.sPzk[zsel], ks_sigma2=ksig,
                           h_of_z_zs=h.h_of_z(z))
    o = hmref.RefHaloModel(ci, z, ks, m, p, skip_nfw=True)
    _, uo = o.add_nfw_profile("nfwnum", numeric=True)
    d = float(np.max(np.abs(u[zsel, msel] - uo)))
    worst = max(worst, d)
    print("rows", zsel, msel, "max |du| =", d)
assert worst < 1e-12, worst
print("ok")
