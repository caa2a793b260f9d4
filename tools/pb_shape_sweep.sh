#!/bin/bash
# Mass-integral launch shapes (HMG_PB_THIN = 0: 128-k tiles, 8 wavefronts; 1: 64-k, 16; 2: 64-k, 8; 3: 128-k, 16) per slab size.
for nz in ${SLABS:-32 16 8 4}; do
  for t in 0 1 2 3; do
    HMG_PB_THIN=$t python bench.py --nz $nz --no-cpu-baseline --no-limber --steps 40 > /tmp/pbs.json 2>/dev/null || { echo "nz=$nz thin=$t failed"; continue; }
    python - $nz $t <<'PY'
import json, sys
d = json.loads(open("/tmp/pbs.json").read().strip().splitlines()[-1])
print(f"nz={sys.argv[1]:>3} thin={sys.argv[2]}  step {d['ms_per_step']:.4f}  power {d['kernels']['power_batch_kernel']['ms']:.4f}")
PY
  done
done
