#!/bin/bash
# Issue-slot accounting of the step's kernels from SQ counters (one rocprofv3 --pmc pass per group).
# Usage: tools/pmc_kernel.sh [kernel-name-substring]
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_sq; mkdir -p $OUT
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $grp -d $OUT/g$i -o p --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-limber --no-readme --no-long-grid --no-graph $PMC_BENCH_FLAGS > $OUT/g$i.log 2>&1 || { tail -5 $OUT/g$i.log; exit 1; }
done
python3 - $OUT "${1:-}" <<'PY'
import csv, glob, re, sys
from collections import defaultdict
out, pat = sys.argv[1], sys.argv[2]
tot, n = defaultdict(lambda: defaultdict(float)), defaultdict(lambda: defaultdict(int))
for f in glob.glob(f"{out}/g*/p_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
import json
sys.path.insert(0, ".")
from hmvec_amd._native import kernel_source_sha16
json.dump({"source_sha16": kernel_source_sha16(), "source": "rocprofv3 --pmc <SQ group> -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-limber --no-readme --no-long-grid --no-graph "
                     "(tools/pmc_kernel.sh); per-launch averages",
           "kernels": {k: {c: tot[k][c] / n[k][c] for c in sorted(tot[k])} for k in sorted(tot)}},
          open(f"{out}/sq_issue_counters.json", "w"), indent=1)
for k in sorted(tot):
    if pat and pat not in k: continue
    print(k)
    for c in sorted(tot[k]):
        print(f"   {c:24s} {tot[k][c]/n[k][c]:16.0f}")
PY
