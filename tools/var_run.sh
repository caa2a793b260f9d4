#!/bin/bash
# Time named build variants hmvec_amd/libhmgrid_<name>.so against each other on one box, alternating.
# Usage: tools/var_run.sh "base mb1 mb2" [rounds] [extra bench flags]
VARS="$1"; R=${2:-2}; shift; shift || true
for i in $(seq $R); do
  for v in $VARS; do
    export HMG_LIB_PATH=$PWD/hmvec_amd/libhmgrid_$v.so
    python bench.py --no-cpu-baseline --no-limber --steps 40 "$@" > /tmp/var_$v.json 2>/tmp/var_$v.err || { tail -3 /tmp/var_$v.err; continue; }
    python - $v <<'PY'
import json, sys
d = json.loads(open(f"/tmp/var_{sys.argv[1]}.json").read().strip().splitlines()[-1])
k = d["kernels"]
print(f"{sys.argv[1]:8s} step {d['ms_per_step']:.4f}  power {k['power_batch_kernel']['ms']:.4f}  nfw {k['nfw_kernel']['ms']:.4f}  fused {k['profile_fused_kernel']['ms']:.4f}")
PY
  done
done
