#!/bin/bash
# LDS bank-conflict counters of the step's kernels (one rocprofv3 --pmc pass).  Usage: tools/pmc_lds.sh [kernel-substring]
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_lds; mkdir -p $OUT
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS -d $OUT/g -o p --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-limber --no-readme --no-long-grid --no-graph $PMC_BENCH_FLAGS > $OUT/g.log 2>&1 || { tail -5 $OUT/g.log; exit 1; }
python3 - $OUT "${1:-}" <<'PY'
import csv, re, sys
from collections import defaultdict
out, pat = sys.argv[1], sys.argv[2]
tot, n = defaultdict(lambda: defaultdict(float)), defaultdict(lambda: defaultdict(int))
for r in csv.DictReader(open(f"{out}/g/p_counter_collection.csv")):
    k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
    tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
for k in sorted(tot):
    if pat and pat not in k: continue
    print(k)
    for c in sorted(tot[k]):
        print(f"   {c:24s} {tot[k][c]/n[k][c]:16.0f}")
PY
