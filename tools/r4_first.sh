#!/bin/bash
# round 4, first GPU call: suite, default bench, and the rocFFT route at the reference callers' radial grid
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/g1_all.log 2>&1 || { tail -30 $O/g1_all.log; exit 1; }
tail -2 $O/g1_all.log
python3 bench.py --no-limber > $O/bench_default.json 2> $O/bench_default.err
echo "default bench done"
python3 bench.py --nxs 30000 --xmax 50 --no-limber --steps 20 --warmup 3 > $O/bench_nxs30000_rocfft.json 2> $O/bench_nxs30000_rocfft.err
echo "nxs30000 bench done"
rocprofv3 --kernel-trace --stats -d $O/kt_nxs30000 -o k --output-format csv -- python3 bench.py --nxs 30000 --xmax 50 --no-cpu-baseline --no-limber --steps 10 --warmup 2 > $O/kt_nxs30000.log 2>&1
cp $O/kt_nxs30000/k_kernel_stats.csv $O/nxs30000_rocfft_kernel_stats.csv
python3 - <<'PY'
import json
for n in ("bench_default","bench_nxs30000_rocfft"):
    d=json.load(open(f"gpurun_out/r4/{n}.json"))
    print(n, d["ms_per_step"], {k:v["ms"] for k,v in d["kernels"].items()}, d.get("cpu_baseline",{}).get("parity_worst_dP_over_tol"))
PY
head -12 $O/nxs30000_rocfft_kernel_stats.csv | cut -c1-160
