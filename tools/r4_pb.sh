#!/bin/bash
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/g4_all.log 2>&1 || { tail -40 $O/g4_all.log; exit 1; }
tail -2 $O/g4_all.log
bash tools/slab_sweep.sh > $O/slab_sweep.txt 2>&1; cat $O/slab_sweep.txt
bash tools/kernel_times.sh "32 4" > $O/kt_pb.txt 2>&1; cat $O/kt_pb.txt
