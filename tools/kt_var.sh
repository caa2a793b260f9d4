#!/bin/bash
# Per-kernel average durations (us) for named build variants, from rocprofv3 kernel stats.
# Usage: tools/kt_var.sh "prev s4" [kernel-substring] [extra bench flags]
VARS="$1"; PAT="${2:-}"; shift; shift || true
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in $VARS; do
  export HMG_LIB_PATH=$PWD/hmvec_amd/libhmgrid_$v.so
  rocprofv3 --kernel-trace --stats -d gpurun_out/ktv_$v -o k --output-format csv -- python3 bench.py --no-cpu-baseline --no-limber --no-readme --no-long-grid --steps 30 "$@" > gpurun_out/ktv_$v.log 2>&1
  python3 - gpurun_out/ktv_$v/k_kernel_stats.csv "$v" "$PAT" <<'PY'
import csv, sys
tot = 0.0
for r in csv.DictReader(open(sys.argv[1])):
    if int(r["Calls"]) < 20: continue
    tot += float(r["AverageNs"]) / 1e3
    if sys.argv[3] and sys.argv[3] not in r["Name"]: continue
    print(f"{sys.argv[2]:8s} {float(r['AverageNs'])/1e3:8.1f} us  {r['Name'][:60]}")
print(f"{sys.argv[2]:8s} {tot:8.1f} us  sum of kernels per step")
PY
done
