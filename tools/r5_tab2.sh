#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python3 - <<'PY' | tee gpurun_out/r5/table_stage.txt
import sys, numpy as np
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import shape_sweep as ss
zs = np.linspace(0.01, 3.0, 32); ms = np.geomspace(2e10, 1e17, 512); ks = np.geomspace(1e-4, 100, 4096)
six = [("nfw", "nfw"), ("electron", "electron"), ("g", "g"), ("nfw", "electron"), ("g", "nfw"), ("g", "electron")]
ss.table_case("table nxs=5000 xmax=20: one-row kernel (table build of the 2500 plan)", zs, ms, ks, 5000, 20.0)
ss.table_case("table nxs=5000 xmax=20: table -> rocFFT chain (HMG_FUSED_FFT=0)", zs, ms, ks, 5000, 20.0, env={"HMG_FUSED_FFT": "0"}, reps=3)
ss.table_case("table nxs=3000 xmax=20: one-row kernel, run-time plan", zs, ms, ks, 3000, 20.0)
ss.table_case("table nxs=30000 xmax=50: long-grid kernel", zs, ms, ks, 30000, 50.0, reps=3)
ss.run_case("nxs=3000 xmax=20: compile-time plan M=1500", zs, ms, ks, (3000, 20), six)
ss.run_case("nxs=3000 xmax=20: run-time plan (HMG_FUSED_GENERIC=1)", zs, ms, ks, (3000, 20), six, env={"HMG_FUSED_GENERIC": "1"})
PY
