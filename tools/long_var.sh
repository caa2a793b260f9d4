#!/bin/bash
# Profile-stage time of the long-grid bench (Config-3 grid, nxs=30000, xmax=50) for named build variants
# (hmvec_amd/libhmgrid_<v>.so; "main" = the working-tree build).  Usage: tools/long_var.sh "main occ6" [rounds]
VARS="$1"; R=${2:-2}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for i in $(seq $R); do
for v in $VARS; do
  if [ $v = main ]; then unset HMG_LIB_PATH; else export HMG_LIB_PATH=$PWD/hmvec_amd/libhmgrid_$v.so; fi
  timeout -k 10 200 python3 bench.py --nxs 30000 --xmax 50 --no-cpu-baseline --no-limber --no-readme --no-long-grid --steps 24 --warmup 3 > /tmp/lv_$v.json 2>/tmp/lv_$v.err || { echo "$v FAILED"; tail -5 /tmp/lv_$v.err; continue; }
  python3 - $v <<'PY'
import json, sys
d = json.loads(open(f"/tmp/lv_{sys.argv[1]}.json").read().strip().splitlines()[-1])
k = d["kernels"]
print(f"{sys.argv[1]:8s} step {d['ms_per_step']:.4f}  profile {k['profile_fused_kernel']['ms']:.4f}  power {k['power_batch_kernel']['ms']:.4f}  nfw {k['nfw_kernel']['ms']:.4f}")
PY
done; done
