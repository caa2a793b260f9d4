# Per-phase VALU / SALU / LDS instruction counts per wavefront of the fused profile kernel: the phase-ablation builds
# (make OUT=../libhmgrid_ablN.so EXTRA=-DHMG_ABL=N, N = 6 scalars only, 1 phase A, 2 +FFT, 3 +unpack) under rocprofv3 --pmc.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in ${ABLS:-abl6 abl1 abl2 abl3 full}; do
  if [ $v = full ]; then unset HMG_LIB_PATH; else export HMG_LIB_PATH=$PWD/hmvec_amd/libhmgrid_$v.so; fi
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS -d gpurun_out/ablpmc_$v -o p --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-limber --no-graph > gpurun_out/ablpmc_$v.log 2>&1
  python3 - gpurun_out/ablpmc_$v/p_counter_collection.csv $v <<'PY'
import csv, sys
from collections import defaultdict
t=defaultdict(float); n=defaultdict(int)
for r in csv.DictReader(open(sys.argv[1])):
    if 'profile_group' not in r['Kernel_Name']: continue
    t[r['Counter_Name']]+=float(r['Counter_Value']); n[r['Counter_Name']]+=1
print(sys.argv[2], {k: round(t[k]/n[k]/131328,1) for k in t})
PY
done
