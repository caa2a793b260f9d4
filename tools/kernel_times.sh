#!/bin/bash
# Per-kernel average durations (us) of the bench step at the given slab sizes, from rocprofv3 kernel stats.
# Usage: tools/kernel_times.sh "32 4"
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for nz in ${1:-32 4}; do
  rocprofv3 --kernel-trace --stats -d gpurun_out/kt_$nz -o k --output-format csv -- python3 bench.py --nz $nz --no-cpu-baseline --no-limber --no-readme --no-long-grid --steps 30 > gpurun_out/kt_$nz.log 2>&1
  echo "== nz=$nz"
  python3 - gpurun_out/kt_$nz/k_kernel_stats.csv <<'PY'
import csv, sys
tot = 0
for r in csv.DictReader(open(sys.argv[1])):
    if int(r["Calls"]) < 20: continue
    us = float(r["AverageNs"]) / 1e3; tot += us
    print(f"{us:8.1f}  {r['Name'][:70]}")
print(f"{tot:8.1f}  sum")
PY
done
