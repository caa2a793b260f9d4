#!/bin/bash
# kernel time of the narrow-band route (Config-3 grid, pressure nxs=30000, xmax=2) for build variants hmvec_amd/libhmgrid_<v>.so
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in ${1:-main}; do
  if [ $v = main ]; then unset HMG_LIB_PATH; else export HMG_LIB_PATH=$PWD/hmvec_amd/libhmgrid_$v.so; fi
  rocprofv3 --kernel-trace --stats -d gpurun_out/r4/bv_$v -o k --output-format csv -- python3 tools/probes/tsz_band_only.py 6 > /dev/null 2>&1
  python3 - $v <<'PY'
import csv, sys
for r in csv.DictReader(open(f"gpurun_out/r4/bv_{sys.argv[1]}/k_kernel_stats.csv")):
    if "band_kernel" in r["Name"]:
        print(f"{sys.argv[1]:6s} {float(r['AverageNs'])/1e6:.3f} ms  {r['Name'][:60]}")
PY
done
