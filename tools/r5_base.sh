#!/bin/bash
# round-5 baseline: GPU suite, headline bench line, long-grid bench line (graph replay), tSZ + numeric-NFW stage probes
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/base_tests.log 2>&1 || { tail -40 $O/base_tests.log; exit 1; }
tail -2 $O/base_tests.log
python bench.py > $O/base_bench.json 2> $O/base_bench.err
python bench.py --nxs 30000 --xmax 50 --no-cpu-baseline --no-limber --no-readme > $O/base_bench_nxs30000.json 2> $O/base_bench_nxs30000.err
python tools/probes/numeric_nfw_routes.py > $O/base_numeric_nfw.txt 2>&1
python tools/probes/tsz_routes.py > $O/base_tsz.txt 2>&1
tail -5 $O/base_numeric_nfw.txt $O/base_tsz.txt
