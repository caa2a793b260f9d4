"""Radial grids between nxs = 5000 and 12288 fit LDS as one row (run-time plan) and are also long grids the pruned
route takes: which is faster?  Config-3 grid, profile stage."""
import sys, numpy as np
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import shape_sweep as ss
zs = np.linspace(0.01, 3.0, 32); ms = np.geomspace(2e10, 1e17, 512); ks = np.geomspace(1e-4, 100, 4096)
six = [("nfw", "nfw"), ("electron", "electron"), ("g", "g"), ("nfw", "electron"), ("g", "nfw"), ("g", "electron")]
for nxs in (6000, 8000, 12000):
    ss.run_case(f"nxs={nxs} xmax=20: one row in LDS, run-time plan", zs, ms, ks, (nxs, 20), six, env={"HMG_FUSED_PREFER_M": "100000"})
    ss.run_case(f"nxs={nxs} xmax=20: long-grid route (the default above M = 2500)", zs, ms, ks, (nxs, 20), six)
