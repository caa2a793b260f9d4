#!/bin/bash
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_longgrid.py -m gpu -x -q > $O/band3_tests.log 2>&1 || { tail -40 $O/band3_tests.log; exit 1; }
tail -2 $O/band3_tests.log
bash tools/long_ab.sh "nopipe main pocc5 pocc4" 2 tsz | tee $O/band3.txt
