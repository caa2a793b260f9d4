cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4/band; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/kt -o k --output-format csv -- python3 tools/probes/tsz_band_only.py 6 > $O/kt.log 2>&1
grep "band_kernel" $O/kt/k_kernel_stats.csv | cut -d, -f1-4 | cut -c1-160
i=0
for grp in "SQ_INSTS_VALU SQ_WAVES SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_ANY" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  rocprofv3 --pmc $grp -d $O/g$i -o p --output-format csv -- python3 tools/probes/tsz_band_only.py 2 > $O/g$i.log 2>&1
done
python3 - $O <<'PY'
import csv, glob, sys
from collections import defaultdict
tot, n = defaultdict(float), defaultdict(int)
for f in glob.glob(sys.argv[1] + "/g*/p_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "band_kernel" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for k in sorted(tot): print(f"{k:24s} {tot[k]/n[k]:16.0f}")
PY
