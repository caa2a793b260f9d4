#!/bin/bash
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/c1_tests.log 2>&1 || { tail -40 $O/c1_tests.log; exit 1; }
tail -2 $O/c1_tests.log
python bench.py --no-cpu-baseline > $O/c1_bench.json 2> $O/c1_bench.err || { tail -20 $O/c1_bench.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5/c1_bench.json").read().strip().splitlines()[-1])
print("step", d["ms_per_step"]); print(json.dumps(d["long_grid"], indent=1)[:2500])
PY
