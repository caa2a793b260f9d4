#!/bin/bash
# Launch-shape sweep for a thin z-slab (default nz=4): NFW tile/threads and mass-integral shape knobs.
# Usage: tools/thin_sweep.sh [nz]
NZ=${1:-4}
run() {
  python bench.py --nz $NZ --no-cpu-baseline --no-limber --steps 60 > /tmp/thin.json 2>/tmp/thin.err || { echo "$1: failed"; tail -2 /tmp/thin.err; return; }
  python - "$1" <<'PY'
import json, sys
d = json.loads(open("/tmp/thin.json").read().strip().splitlines()[-1])
k = d["kernels"]
print(f"{sys.argv[1]:32s} step {d['ms_per_step']:.4f}  power {k['power_batch_kernel']['ms']:.4f}  nfw {k['nfw_kernel']['ms']:.4f}  fused {k['profile_fused_kernel']['ms']:.4f}")
PY
}
for rep in 1 2; do
run "default"
HMG_NFW_THREADS=128 run "nfw thr=128 ktile=4096"
HMG_NFW_THREADS=64 run "nfw thr=64 ktile=4096"
HMG_NFW_THREADS=128 HMG_NFW_KTILE=8192 run "nfw thr=128 ktile=8192"
done
