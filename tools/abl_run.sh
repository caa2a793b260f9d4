#!/bin/bash
# Phase ablation of the fused profile kernel (timing only; results of the ablated builds are garbage):
# libhmgrid_abl{1,2,3}.so stop after phase A / B / C (make OUT=../libhmgrid_ablN.so EXTRA=-DHMG_ABL=N).
for v in ${ABLS:-abl1 abl2 abl3 full}; do
  if [ $v = full ]; then unset HMG_LIB_PATH; else export HMG_LIB_PATH=$PWD/hmvec_amd/libhmgrid_$v.so; fi
  python bench.py --no-cpu-baseline --no-limber --steps 40 "$@" > /tmp/abl_$v.json 2>/tmp/abl_$v.err || { tail -3 /tmp/abl_$v.err; continue; }
  python - $v <<'PY'
import json, sys
d = json.loads(open(f"/tmp/abl_{sys.argv[1]}.json").read().strip().splitlines()[-1])
print(f"{sys.argv[1]:5s} fused {d['kernels']['profile_fused_kernel']['ms']:.4f}  step {d['ms_per_step']:.4f}")
PY
done
