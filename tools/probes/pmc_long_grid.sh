#!/bin/bash
# HBM traffic of the long-grid kernel (Config-3 grid, nxs=30000, xmax=50): FETCH_SIZE / WRITE_SIZE in separate passes,
# bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH_SIZE correction, MI355X_MICROARCH.md)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4/pmc_long; mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d $O/$c -o p --output-format csv -- python3 bench.py --nxs 30000 --xmax 50 --steps 3 --warmup 1 --no-cpu-baseline --no-limber --no-readme --no-long-grid --no-graph > $O/$c.log 2>&1
done
python3 - $O <<'PY'
import csv, re, sys
from collections import defaultdict
o = sys.argv[1]
def per(path, counter):
    tot, n = defaultdict(float), defaultdict(int)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            k = re.sub(r"\(.*", "", r["Kernel_Name"]); tot[k] += float(r["Counter_Value"]); n[k] += 1
    return {k: tot[k] / n[k] for k in tot}
f = per(f"{o}/FETCH_SIZE/p_counter_collection.csv", "FETCH_SIZE"); w = per(f"{o}/WRITE_SIZE/p_counter_collection.csv", "WRITE_SIZE")
for k in sorted(set(f) | set(w)):
    b = (2 * f.get(k, 0) + w.get(k, 0)) * 1024
    if b > 5e6: print(f"{b/1e6:9.1f} MB  (fetch {f.get(k,0)/1e3:8.1f} MB counted, write {w.get(k,0)/1e3:8.1f} MB)  {k[:70]}")
PY
