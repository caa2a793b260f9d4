#!/bin/bash
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/g3_all.log 2>&1 || { tail -40 $O/g3_all.log; exit 1; }
tail -2 $O/g3_all.log
timeout -k 10 600 python3 bench.py > $O/bench_default2.json 2> $O/bench_default2.err || { tail -20 $O/bench_default2.err; exit 1; }
python3 - <<'PY'
import json
d=json.load(open("gpurun_out/r4/bench_default2.json"))
print("step", d["ms_per_step"], "launches", d["launches_per_step"], d["pcie"]["streamed"], d["pcie"]["h2d_inputs_ms"], d["pcie"]["h2d_inputs_pinned_ms"], d["pcie"]["d2h_results_ms"])
print(json.dumps(d["readme_config2"], indent=1))
PY
