#!/bin/bash
# Kernel-stats CSV + SQ / LDS / memory counters of the kernels of ONE python command, per-launch averages, for the
# kernels whose name contains a substring.  Separate rocprofv3 passes: --kernel-trace --stats first, then one --pmc pass
# per counter group (gpurun refuses --pmc combined with traces; the program stands directly behind `--`).
# Usage (GPU box): tools/kernel_counters.sh <outdir> <kernel-substring> <script.py> [args...]
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=$1; PAT=$2; shift; shift
mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/kt -o k --output-format csv -- python3 "$@" > $OUT/kt.log 2>&1 || { tail -5 $OUT/kt.log; exit 1; }
cp $OUT/kt/k_kernel_stats.csv $OUT/kernel_stats.csv
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" \
           "FETCH_SIZE" "WRITE_SIZE" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $grp -d $OUT/g$i -o p --output-format csv -- python3 "$@" > $OUT/g$i.log 2>&1 || { echo "group '$grp' failed"; tail -3 $OUT/g$i.log; }
done
python3 - $OUT "$PAT" <<'PY'
import csv, glob, re, sys
from collections import defaultdict
out, pat = sys.argv[1], sys.argv[2]
tot, n = defaultdict(lambda: defaultdict(float)), defaultdict(lambda: defaultdict(int))
for f in glob.glob(f"{out}/g*/p_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
        if pat in k:
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
stats = {}
for r in csv.DictReader(open(f"{out}/kernel_stats.csv")):
    k = re.sub(r"\(.*", "", r["Name"]).replace("void ", "")
    if pat in k:
        stats[k] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3)
with open(f"{out}/summary.txt", "w") as fo:
    for k in sorted(tot):
        c = {m: tot[k][m] / n[k][m] for m in tot[k]}
        calls, us = stats.get(k, (0, float("nan")))
        lines = [f"{k}", f"   rocprofv3 --kernel-trace --stats: {calls} calls, average {us:.1f} us"]
        for m in sorted(c):
            lines.append(f"   {m:24s} {c[m]:16.0f}")
        if "SQ_INSTS_VALU" in c:
            issue_us = c["SQ_INSTS_VALU"] * 4 / 1024 / 2.4e3
            lines.append(f"   -> VALU issue bound (x 4 cycles / 1024 SIMDs / 2.4 GHz): {issue_us:.1f} us = {issue_us / us:.2f} of the launch")
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            b = (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024
            lines.append(f"   -> HBM bytes (2 FETCH_SIZE + WRITE_SIZE) x 1024: {b / 1e6:.1f} MB = {b / us / 1e6:.2f} TB/s")
        if "SQ_LDS_IDX_ACTIVE" in c:
            lines.append(f"   -> LDS array active {c['SQ_LDS_IDX_ACTIVE'] / 256 / 2.4e3:.1f} us per CU ({c['SQ_LDS_IDX_ACTIVE'] / 256 / 2.4e3 / us:.2f} of the launch), "
                         f"bank-conflict cycles {c.get('SQ_LDS_BANK_CONFLICT', 0) / max(c['SQ_LDS_IDX_ACTIVE'], 1):.2f} of them")
        if "SQ_WAVE_CYCLES" in c:
            wc = c["SQ_WAVE_CYCLES"]
            lines.append(f"   -> of the wave-cycles: waiting (barrier/memory) {c.get('SQ_WAIT_ANY', 0) / wc:.2f}, issue-stalled {c.get('SQ_WAIT_INST_ANY', 0) / wc:.2f}, "
                         f"issuing {c.get('SQ_ACTIVE_INST_ANY', 0) / wc:.2f}; mean occupancy {wc * 4 / (us * 2.4e3 * 1024):.1f} waves/SIMD")
        print("\n".join(lines)); fo.write("\n".join(lines) + "\n")
PY
