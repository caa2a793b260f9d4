#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5; mkdir -p $O
echo "# HMG_CHIRP=0: every row takes the decomposition; abl bits: 4 no LDS passes, 8 no unpack, 16 no mode loads in the interpolation, 32 no group loop, 64 no transcendentals" | tee $O/abl2.txt
HMG_CHIRP=0 bash tools/long_ab.sh "nolpt abl12 abl28 abl44 abl108 abl124" 2 gas | tee -a $O/abl2.txt
