#!/bin/bash
# VERDICT r03 #3 "measure both": the per-z chain as a role of the profile group (default) against a launch of its own in
# front of the stand-alone row kernel (HMG_X=chain_alone: 5 launches per pass, no private segment in the row kernel).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for nz in 32 4; do for i in 1 2; do for x in "" chain_alone; do
  HMG_X=$x python3 bench.py --nz $nz --no-cpu-baseline --no-limber --no-readme --no-long-grid --steps 40 > /tmp/ca.json 2>/dev/null
  python3 - "$x" $nz <<'PY'
import json, sys
d = json.loads(open("/tmp/ca.json").read().strip().splitlines()[-1])
print(f"nz={sys.argv[2]:>2} {sys.argv[1] or 'grouped':12s} step {d['ms_per_step']:.4f}  launches {d['launches_per_step']}  profile stage {d['kernels']['profile_fused_kernel']['ms']:.4f}")
PY
done; done; done
