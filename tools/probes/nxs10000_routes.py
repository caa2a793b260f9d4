import sys, numpy as np
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import shape_sweep as ss
zs = np.linspace(0.01, 3.0, 32); ms = np.geomspace(2e10, 1e17, 512); ks = np.geomspace(1e-4, 100, 4096)
six = [("nfw", "nfw"), ("electron", "electron"), ("g", "g"), ("nfw", "electron"), ("g", "nfw"), ("g", "electron")]
ss.run_case("nxs=10000 xmax=20: one row in LDS, compile-time plan M=5000 (80 KB)", zs, ms, ks, (10000, 20), six)
ss.run_case("nxs=10000 xmax=20: long-grid route (HMG_FUSED_MAX_M=2500)", zs, ms, ks, (10000, 20), six, env={"HMG_FUSED_MAX_M": "2500"})
ss.run_case("nxs=10000 xmax=20: rocFFT route (HMG_FUSED_MAX_M=2500 HMG_PRUNED_FFT=0)", zs, ms, ks, (10000, 20), six, env={"HMG_FUSED_MAX_M": "2500", "HMG_PRUNED_FFT": "0"}, reps=5)
