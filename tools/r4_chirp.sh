#!/bin/bash
# long-grid route: tests, then profile stage / step for the chirp window settings (and build variants)
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_longgrid.py tests/test_gpu_edges.py -m gpu -x -q > $O/g6_long.log 2>&1 || { tail -40 $O/g6_long.log; exit 1; }
tail -2 $O/g6_long.log
for lib in ${LIBS:-main}; do
for nw in ${WINDOWS:-2 1 0}; do
  if [ $lib = main ]; then unset HMG_LIB_PATH; else export HMG_LIB_PATH=$PWD/hmvec_amd/libhmgrid_$lib.so; fi
  HMG_CHIRP_WINDOWS=$nw timeout -k 10 300 python3 bench.py --nxs 30000 --xmax 50 --no-cpu-baseline --no-limber --no-readme --no-long-grid --steps 24 --warmup 3 > /tmp/ch_$nw.json 2>/tmp/ch_$nw.err || { tail -20 /tmp/ch_$nw.err; exit 1; }
  python3 - $nw $lib <<'PY'
import json, sys
d = json.loads(open(f"/tmp/ch_{sys.argv[1]}.json").read().strip().splitlines()[-1])
print(f"{sys.argv[2]} HMG_CHIRP_WINDOWS={sys.argv[1]} step {d['ms_per_step']:.4f}  profile {d['kernels']['profile_fused_kernel']['ms']:.4f}")
PY
done; done
