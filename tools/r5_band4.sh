#!/bin/bash
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_longgrid.py -m gpu -x -q > $O/band4_tests.log 2>&1 || { tail -40 $O/band4_tests.log; exit 1; }
tail -2 $O/band4_tests.log
bash tools/long_ab.sh "base main" 3 tsz | tee $O/band4.txt
