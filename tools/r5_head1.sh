#!/bin/bash
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/head1_tests.log 2>&1 || { tail -40 $O/head1_tests.log; exit 1; }
tail -2 $O/head1_tests.log
bash tools/ab_run.sh 3 | tee $O/head1_ab.txt
