#!/bin/bash
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/g5_all.log 2>&1 || { tail -40 $O/g5_all.log; exit 1; }
tail -2 $O/g5_all.log
timeout -k 10 900 python3 tools/shape_sweep.py > $O/shape_sweep.txt 2>$O/shape_sweep.err || { tail -20 $O/shape_sweep.err; exit 1; }
cat $O/shape_sweep.txt
for v in rt8 rt6; do
  echo "== run-time plan compiled for ${v#rt} waves/SIMD"
  HMG_LIB_PATH=$PWD/hmvec_amd/libhmgrid_$v.so python3 - <<'PY'
import sys, numpy as np
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import shape_sweep as ss
zs = np.linspace(0.01, 3.0, 32); ms = np.geomspace(2e10, 1e17, 512); ks = np.geomspace(1e-4, 100, 4096)
six = [("nfw", "nfw"), ("electron", "electron"), ("g", "g"), ("nfw", "electron"), ("g", "nfw"), ("g", "electron")]
ss.run_case("nxs=3000 xmax=20: run-time plan", zs, ms, ks, (3000, 20), six)
ss.run_case("nxs=2000 xmax=20: run-time plan (HMG_FUSED_GENERIC=1)", zs, ms, ks, (2000, 20), six, env={"HMG_FUSED_GENERIC": "1"})
PY
done
