#!/bin/bash
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_longgrid.py -m gpu -x -q > $O/long2_tests.log 2>&1 || { tail -40 $O/long2_tests.log; exit 1; }
tail -2 $O/long2_tests.log
bash tools/long_ab.sh "base main" 2 | tee $O/long2_ab.txt
echo "# HMG_CHIRP=0" | tee -a $O/long2_ab.txt
HMG_CHIRP=0 bash tools/long_ab.sh "base main abl12 abl44" 1 gas | tee -a $O/long2_ab.txt
