#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5; mkdir -p $O
bash tools/long_var.sh "base main" 2 | tee $O/long3_var.txt
echo "# HMG_CHIRP=0 (all rows: decomposition); abl 4 = no LDS passes, 8 = no unpack" | tee -a $O/long3_var.txt
HMG_CHIRP=0 bash tools/long_ab.sh "main abl4 abl8 abl12" 2 gas | tee -a $O/long3_var.txt
