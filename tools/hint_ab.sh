#!/bin/bash
# Same box, alternating: constant-prefix hints off (HMG_NO_HINTS=1) / on.
set -e
for i in 1 2 3; do
  for v in 1 0; do
    HMG_NO_HINTS=$v python bench.py --no-cpu-baseline --no-limber --steps 100 > /tmp/h_$v.json
    python - $v <<'PY'
import json, sys
d = json.loads(open(f"/tmp/h_{sys.argv[1]}.json").read().strip().splitlines()[-1])
print(f"no_hints={sys.argv[1]} step {d['ms_per_step']:.4f}  power {d["kernels"]["power_batch_kernel"]["ms"]:.4f}  fused {d["kernels"]["profile_fused_kernel"]["ms"]:.4f}")
PY
  done
done
