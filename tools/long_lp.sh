#!/bin/bash
# Profile-stage time of the long-grid bench for each sub-transform length LP the grid admits (HMG_PRUNED_LP_MIN).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for lp in ${1:-1000 1250 1500 2500}; do
  HMG_PRUNED_LP_MIN=$lp timeout -k 10 200 python3 bench.py --nxs ${NXS:-30000} --xmax ${XMAX:-50} --no-cpu-baseline --no-limber --no-readme --no-long-grid --steps 24 --warmup 3 > /tmp/lp_$lp.json 2>/tmp/lp_$lp.err || { echo "$lp FAILED"; tail -5 /tmp/lp_$lp.err; continue; }
  python3 - $lp <<'PY'
import json, sys
d = json.loads(open(f"/tmp/lp_{sys.argv[1]}.json").read().strip().splitlines()[-1])
print(f"LP>={sys.argv[1]:6s} step {d['ms_per_step']:.4f}  profile {d['kernels']['profile_fused_kernel']['ms']:.4f}")
PY
done
