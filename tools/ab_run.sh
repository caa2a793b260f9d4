#!/bin/bash
# A/B on one box: alternates the committed build (libhmgrid_base.so) and the working-tree build.
# Usage: tools/ab_run.sh [rounds] [extra bench flags]
set -e
R=${1:-3}; shift || true
for i in $(seq $R); do
  for v in base new; do
    if [ $v = base ]; then export HMG_LIB_PATH=$PWD/hmvec_amd/libhmgrid_base.so; else unset HMG_LIB_PATH; fi
    python bench.py --no-cpu-baseline --no-limber --no-readme --no-long-grid --steps 40 "$@" > /tmp/ab_$v.json
    python - $v <<'PY'
import json, sys
d = json.loads(open(f"/tmp/ab_{sys.argv[1]}.json").read().strip().splitlines()[-1])
k = d["kernels"]
f = lambda x: "   -  " if x is None else f"{x:.4f}"      # (tensor group: the NFW rows have no launch of their own)
print(f"{sys.argv[1]:5s} step {d['ms_per_step']:.4f}  power {f(k['power_batch_kernel']['ms'])}  nfw {f(k['nfw_kernel']['ms'])}  fused {f(k['profile_fused_kernel']['ms'])}  host_issue {d['host_issue_ms_per_step']:.4f}")
PY
  done
done
