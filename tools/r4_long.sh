#!/bin/bash
# long-grid route: tests, then the bench at the reference callers' radial grid
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_longgrid.py tests/test_gpu_edges.py -m gpu -x -q > $O/g2_long.log 2>&1 || { tail -40 $O/g2_long.log; exit 1; }
tail -2 $O/g2_long.log
timeout -k 10 300 python3 bench.py --nxs 30000 --xmax 50 --no-limber --steps 20 --warmup 3 > $O/bench_nxs30000_pruned.json 2> $O/bench_nxs30000_pruned.err || { tail -20 $O/bench_nxs30000_pruned.err; exit 1; }
python3 - <<'PY'
import json
for n in ("bench_nxs30000_pruned",):
    d=json.load(open(f"gpurun_out/r4/{n}.json"))
    print(n, d["ms_per_step"], {k:v["ms"] for k,v in d["kernels"].items()}, d.get("cpu_baseline",{}).get("parity_worst_dP_over_tol"))
PY
