#!/bin/bash
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/tab_tests.log 2>&1 || { tail -60 $O/tab_tests.log; exit 1; }
tail -2 $O/tab_tests.log
bash tools/long_ab.sh "base main" 2 | tee $O/tab_long_ab.txt
