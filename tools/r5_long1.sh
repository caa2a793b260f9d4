#!/bin/bash
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_longgrid.py -m gpu -x -q > $O/long1_tests.log 2>&1 || { tail -40 $O/long1_tests.log; exit 1; }
tail -2 $O/long1_tests.log
HMG_LIB_PATH=$PWD/hmvec_amd/libhmgrid_nt256.so timeout -k 10 600 python -m pytest tests/test_gpu_longgrid.py -m gpu -x -q > $O/long1_tests_nt256.log 2>&1 || { tail -30 $O/long1_tests_nt256.log; echo NT256 FAILED; }
tail -2 $O/long1_tests_nt256.log
bash tools/long_ab.sh "base main nolpt noulds noord licm nt256" 2 | tee $O/long1_ab.txt
