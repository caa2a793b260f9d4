"""cProfile of the README usage sequence (20 x 200 x 1001) on the GPU path: where does the host time go?"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import hmvec_amd as hm

zs = np.linspace(0., 3., 20); ms = np.geomspace(2e10, 1e17, 200); ks = np.geomspace(1e-4, 100, 1001)
PAIRS = [("nfw", "nfw"), ("electron", "electron"), ("g", "g"), ("nfw", "electron"), ("g", "nfw"), ("g", "electron")]

def run():
    h = hm.HaloModel(zs, ks, ms=ms, accuracy="low", engine="analytic")
    h.add_battaglia_profile("electron", family="AGN", xmax=20, nxs=5000)
    h.add_hod("g", mthresh=10 ** 10.5 + zs * 0.)
    return {(a, b): h.get_power_1halo(a, b) + h.get_power_2halo(a, b) for a, b in PAIRS}

for _ in range(3):
    run()
t0 = time.perf_counter()
for _ in range(50):
    run()
print("ms per README sequence:", (time.perf_counter() - t0) / 50 * 1e3)
pr = cProfile.Profile(); pr.enable()
for _ in range(50):
    run()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
