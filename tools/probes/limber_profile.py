"""Where does the host time of C_kk + C_kg on device-resident spectra go?  (cProfile, 300 repetitions)"""
import cProfile, os, pstats, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import hmvec_amd as hm

zs = np.linspace(0.01, 3.0, 32); ms = np.geomspace(2e10, 1e17, 512); ks = np.geomspace(1e-4, 100, 4096)
h = hm.HaloModel(zs, ks, ms=ms, accuracy="low", engine="analytic")
h.add_hod("g", mthresh=10 ** 10.5 + zs * 0.0)
blk = h.spectra_block([("nfw", "nfw"), ("g", "nfw")])
blk.compute()
ells = np.linspace(100, 6000, 2000)
def pair(p):
    i = blk.pairs.index(p)
    return (blk.views[2 * i], blk.views[2 * i + 1])
def once():
    a = h.C_kk(ells, zs, ks, pair(("nfw", "nfw")), lzs1=2.5, lzs2=2.5)
    b = h.C_kg(ells, zs, ks, pair(("g", "nfw")), gzs=0.8, lzs=2.5)
    return a, b
for _ in range(20):
    once()
t0 = time.perf_counter()
for _ in range(300):
    once()
print("ms per (C_kk, C_kg):", (time.perf_counter() - t0) / 300 * 1e3)
pr = cProfile.Profile(); pr.enable()
for _ in range(300):
    once()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
