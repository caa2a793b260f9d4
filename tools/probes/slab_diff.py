"""Which stage of the Battaglia path differs between a z-slab model and the full grid?"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import hmvec_amd as hm
zs = np.linspace(0.1, 2.6, 8); ms = np.geomspace(2e10, 1e16, 96); ks = np.geomspace(1e-3, 50, 384)
def build(z):
    h = hm.HaloModel(z, ks, ms=ms, accuracy="low", engine="analytic")
    h.add_battaglia_profile("electron", family="AGN", xmax=20, nxs=1000)
    h.add_hod("g", mthresh=10 ** (10.3 + 0.1 * z))
    return h
full, slab = build(zs), build(zs[4:8])
nzf, nzs, nm = 8, 4, ms.size
def cmp(name, a, b):
    a, b = np.asarray(a), np.asarray(b)
    eq = np.array_equal(a, b)
    print(f"{name:28s} equal={eq}" + ("" if eq else f"  max rel {np.nanmax(np.abs(a-b)/np.maximum(np.abs(b),1e-300)):.2e}  n_diff={int(np.sum(a!=b))}"))
for k in ("m200c", "r200c"):
    cmp(k, full._buf(k, (nzf, nm)).numpy()[4:8], slab._buf(k, (nzs, nm)).numpy())
for k in ("_d_cs", "_d_rvir", "_d_rs"):
    cmp(k, getattr(full, k).numpy()[4:8], getattr(slab, k).numpy())
key = [k for k in full._pool if isinstance(k, tuple) and "rowp" in k]
print("rowp keys", key[:8])
for k in key:
    cmp(str(k), full._pool[k].numpy().reshape(nzf, -1)[4:8], slab._pool[k].numpy().reshape(nzs, -1))
cmp("uk electron", full.uk_profiles["electron"][4:8], slab.uk_profiles["electron"])
cmp("uk nfw", full.uk_profiles["nfw"][4:8], slab.uk_profiles["nfw"])
for a, b in (("electron", "electron"), ("nfw", "electron"), ("g", "g")):
    cmp(f"P1h {a},{b}", full.get_power_1halo(a, b)[4:8], slab.get_power_1halo(a, b))
    cmp(f"P2h {a},{b}", full.get_power_2halo(a, b)[4:8], slab.get_power_2halo(a, b))
