#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5; mkdir -p $O
echo skip-tests

for rep in 1 2 3; do
for x in "" "mix"; do
for nxs in "5000 20" "30000 50"; do
  set -- $nxs
  HMG_X="$x" timeout -k 10 300 python3 bench.py --nxs $1 --xmax $2 --no-cpu-baseline --no-limber --no-readme --no-long-grid --steps 40 --warmup 5 > /tmp/lane.json 2>/tmp/lane.err || { echo "FAILED x=$x nxs=$1"; tail -5 /tmp/lane.err; continue; }
  python3 - "$x" $1 <<'PY'
import json, sys
d = json.loads(open("/tmp/lane.json").read().strip().splitlines()[-1])
k = d["kernels"]
print(f"HMG_X={sys.argv[1]:6s} nxs={sys.argv[2]:6s} step {d['ms_per_step']:.4f}  profile {k['profile_fused_kernel']['ms']}  power {k['power_batch_kernel']['ms']:.4f}  nfw {k['nfw_kernel']['ms']} launches {d['launches_per_step']}", flush=True)
PY
done; done; done | tee $O/mix_experiment.txt
