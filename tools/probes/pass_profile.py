"""Where does the host time of one facade pass at Config 3 go?  (cProfile over 300 passes; the GPU work is 0.47 ms per pass,
the facade adds what this prints.)"""
import cProfile, os, pstats, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import hmvec_amd as hm

zs = np.linspace(0.01, 3.0, 32); ms = np.geomspace(2e10, 1e17, 512); ks = np.geomspace(1e-4, 100, 4096)
PAIRS = [("nfw", "nfw"), ("electron", "electron"), ("g", "g"), ("nfw", "electron"), ("g", "nfw"), ("g", "electron")]
h = hm.HaloModel(zs, ks, ms=ms, accuracy="low", engine="analytic")
thr = 10 ** 10.5 + zs * 0.0
h.add_battaglia_profile("electron", family="AGN", xmax=20, nxs=5000)
h.add_hod("g", mthresh=thr)
blk = h.spectra_block(PAIRS)
def one_pass():
    h.init_mass_function(ms)
    h.add_nfw_profile("nfw", ignore_existing=True)
    h.add_battaglia_profile("electron", family="AGN", xmax=20, nxs=5000, ignore_existing=True)
    h.add_hod("g", mthresh=thr, ignore_existing=True)
    blk.compute()
for _ in range(20):
    one_pass()
h._ctx().sync()
t0 = time.perf_counter()
for _ in range(300):
    one_pass()
t_issue = time.perf_counter() - t0
h._ctx().sync()
print("ms per pass: host issue", t_issue / 300 * 1e3, " wall", (time.perf_counter() - t0) / 300 * 1e3)
pr = cProfile.Profile(); pr.enable()
for _ in range(300):
    one_pass()
pr.disable()
h._ctx().sync()
pstats.Stats(pr).sort_stats("tottime").print_stats(30)
