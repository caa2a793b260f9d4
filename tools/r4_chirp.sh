#!/bin/bash
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_longgrid.py tests/test_gpu_edges.py -m gpu -x -q > $O/g6_long.log 2>&1 || { tail -40 $O/g6_long.log; exit 1; }
tail -2 $O/g6_long.log
for ch in 1 0; do
  HMG_CHIRP=$ch timeout -k 10 300 python3 bench.py --nxs 30000 --xmax 50 --no-cpu-baseline --no-limber --no-readme --steps 24 --warmup 3 > /tmp/ch_$ch.json 2>/tmp/ch_$ch.err || { tail -20 /tmp/ch_$ch.err; exit 1; }
  python3 - $ch <<'PY'
import json, sys
d = json.loads(open(f"/tmp/ch_{sys.argv[1]}.json").read().strip().splitlines()[-1])
print(f"HMG_CHIRP={sys.argv[1]} step {d['ms_per_step']:.4f}  profile {d['kernels']['profile_fused_kernel']['ms']:.4f}")
PY
done
