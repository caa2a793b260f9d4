"""Wall time of the README usage sequence (BASELINE configs[0]/[1]: 20 x 200 x 1001) on the GPU
path, host work included, vs the CPU oracle on the same box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hmvec_amd as hm
from hmvec_amd.params import battaglia_defaults, default_params
from oracle import hmref

zs = np.linspace(0., 3., 20); ms = np.geomspace(2e10, 1e17, 200); ks = np.geomspace(1e-4, 100, 1001)
PAIRS = [("nfw", "nfw"), ("electron", "electron"), ("g", "g"), ("nfw", "electron"), ("g", "nfw"), ("g", "electron")]

def gpu_run():
    t0 = time.perf_counter()
    h = hm.HaloModel(zs, ks, ms=ms, accuracy="low", engine="analytic")
    t1 = time.perf_counter()
    h.add_battaglia_profile("electron", family="AGN", xmax=20, nxs=5000)
    h.add_hod("g", mthresh=10 ** 10.5 + zs * 0.)
    out = {}
    for a, b in PAIRS:
        out[(a, b)] = h.get_power_1halo(a, b) + h.get_power_2halo(a, b)
    t2 = time.perf_counter()
    return h, out, (t1 - t0, t2 - t1)

gpu_run()                                   # warm-up: library load, context, first launches
h, out, (tc, tr) = gpu_run()
print(f"GPU path: ctor {tc*1e3:.2f} ms, add_battaglia+add_hod+12 get_power_* calls {tr*1e3:.2f} ms")

p = dict(default_params)
ksig = np.geomspace(p["sigma2_kmin"], p["sigma2_kmax"], p["sigma2_numks"])
t0 = time.perf_counter()
ci = hmref.CosmoInputs(h=h.h, omm0=h.omm0, ombh2=p["ombh2"], rho_crit_0=float(h.rho_critical_z(0.0)),
                       rho_crit_zs=h.rho_critical_z(zs), Pzk=h.Pzk, sPzk=h.sPzk, ks_sigma2=ksig,
                       h_of_z_zs=h.h_of_z(zs))
o = hmref.RefHaloModel(ci, zs, ks, ms, p)
t1 = time.perf_counter()
o.add_battaglia_profile("electron", "AGN", p["battaglia_gas_gamma"], battaglia_defaults["AGN"], 5000, 20)
o.add_hod("g", mthresh=10 ** 10.5 + zs * 0.)
ref = {pr: o.get_power_1halo(*pr) + o.get_power_2halo(*pr) for pr in PAIRS}
t2 = time.perf_counter()
print(f"CPU oracle: ctor {(t1-t0)*1e3:.0f} ms, rest {(t2-t1)*1e3:.0f} ms")
worst = max(float(np.max(np.abs(out[k] - ref[k]) / (1e-8 * np.abs(ref[k]) + 1e-12 * np.max(np.abs(ref[k]), axis=-1, keepdims=True)))) for k in PAIRS)
print("worst |dP|/tol =", worst)
