#!/bin/bash
# Per-rank proxy for the z-slab scaling curve: one GPU, slabs of nz = 32/16/8/4 (what a rank of a
# 1/2/4/8-GPU job computes).  Usage: tools/slab_sweep.sh [extra bench.py flags]; env passes through.
set -e
for nz in ${SLABS:-32 16 8 4}; do
  python bench.py --nz $nz --no-cpu-baseline --no-limber --no-readme --no-long-grid --steps 50 "$@" > /tmp/slab_$nz.json
  python - $nz <<'PY'
import json, sys
d = json.loads(open(f"/tmp/slab_{sys.argv[1]}.json").read().strip().splitlines()[-1])
k = d["kernels"]
f = lambda x: "   -  " if x is None else f"{x:.4f}"      # (tensor group: the NFW rows have no launch of their own)
print(f"nz={sys.argv[1]:>3}  ms_per_step={d['ms_per_step']:.4f}  host_issue_ms={d['host_issue_ms_per_step']:.4f}  "
      f"power={f(k['power_batch_kernel']['ms'])} nfw={f(k['nfw_kernel']['ms'])} fused={f(k['profile_fused_kernel']['ms'])}  [{d['launch_mode']}]")
PY
done
