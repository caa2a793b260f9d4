#!/bin/bash
# Regenerates the judged profile artefacts of a round on the GPU box (run through gpurun):
#   gpurun_out/prof/config3_bench.json          bench.py JSON line (default run, with cpu_baseline)
#   gpurun_out/prof/config3_kernel_stats.csv    rocprofv3 --kernel-trace --stats of the same command
#   gpurun_out/prof/pmc_{fetch,write}_counter_collection.csv + pmc_traffic.json
#   gpurun_out/prof/sq_issue_counters.{json,txt}  (tools/pmc_kernel.sh)
# The counter passes run first and their summaries are installed under profiles/$ROUND/ on the box before the
# bench line is taken, so the line's roofline numbers come from this build's own counters.
# Copy them into profiles/rNN/ afterwards.  PMC passes run separately from the trace (gpurun rule).
set -e
ROOT="$GRAFT_REPO_ROOT"; [ -z "$ROOT" ] && ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
OUT=gpurun_out/prof; mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d $OUT/pmc_$c -o p --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-limber --no-readme --no-long-grid --no-graph > $OUT/pmc_$c.log 2>&1
done
cp $OUT/pmc_FETCH_SIZE/p_counter_collection.csv $OUT/pmc_fetch_counter_collection.csv
cp $OUT/pmc_WRITE_SIZE/p_counter_collection.csv $OUT/pmc_write_counter_collection.csv
python3 - $OUT <<'PY'
import csv, json, re, sys
from collections import defaultdict
out = sys.argv[1]
def per_kernel(path, counter):
    tot, n = defaultdict(float), defaultdict(int)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = re.sub(r"\(.*", "", r["Kernel_Name"])
        tot[name] += float(r["Counter_Value"]); n[name] += 1
    return {k: tot[k] / n[k] for k in tot}
cnt = defaultdict(int)
for r in csv.DictReader(open(f"{out}/pmc_fetch_counter_collection.csv")):
    if r["Counter_Name"] == "FETCH_SIZE":
        cnt[re.sub(r"\(.*", "", r["Kernel_Name"])] += 1
steps = max([v for k, v in cnt.items() if "power_batch_kernel" in k] + [1])     # one mass-integral launch per step
f = per_kernel(f"{out}/pmc_fetch_counter_collection.csv", "FETCH_SIZE")
w = per_kernel(f"{out}/pmc_write_counter_collection.csv", "WRITE_SIZE")
sys.path.insert(0, ".")
from hmvec_amd._native import kernel_source_sha16
res = {"source_sha16": kernel_source_sha16(),
       "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --steps 3 --warmup 1 "
                 "--no-cpu-baseline --no-limber --no-readme --no-long-grid --no-graph; Config 3, 1 GPU (tools/profile_round.sh)",
       "correction": "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH_SIZE counts 128-B requests as 64 B; "
                     "MI355X_MICROARCH.md, HBM section)",
       "kernels": {k: {"FETCH_SIZE_KB_per_launch": f.get(k, 0.0), "WRITE_SIZE_KB_per_launch": w.get(k, 0.0),
                       "hbm_bytes_per_launch_corrected": (2 * f.get(k, 0.0) + w.get(k, 0.0)) * 1024,
                       "launches_per_step": round(cnt.get(k, 0) / max(steps, 1), 3)} for k in sorted(set(f) | set(w))}}
json.dump(res, open(f"{out}/pmc_traffic.json", "w"), indent=1)
for k, v in res["kernels"].items():
    print(f"{v['hbm_bytes_per_launch_corrected']/1e6:10.1f} MB  {k}")
PY
# SQ issue counters (five more passes), then install both summaries where bench.py reads them, so that the
# bench line below is derived from the counters of THIS build on THIS box
ROUND=${ROUND:-r06}
bash tools/pmc_kernel.sh > $OUT/sq_issue_counters.txt
cp gpurun_out/pmc_sq/sq_issue_counters.json $OUT/sq_issue_counters.json
mkdir -p profiles/$ROUND
cp $OUT/pmc_traffic.json $OUT/sq_issue_counters.json profiles/$ROUND/
echo "counters done"
python3 bench.py > $OUT/config3_bench.json
echo "bench done"
rocprofv3 --kernel-trace --stats -d $OUT/kt -o k --output-format csv -- python3 bench.py --no-cpu-baseline --no-limber --no-readme --no-long-grid > $OUT/kt.log 2>&1
cp $OUT/kt/k_kernel_stats.csv $OUT/config3_kernel_stats.csv
echo "kernel trace done"
