#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5; mkdir -p $O
echo "# tSZ band kernel ablations: 4 = no LDS passes, 8 = no accumulation, 64 = no transcendentals, 76 = all three" | tee $O/band1.txt
bash tools/long_ab.sh "main abl4 abl8 abl64 abl76" 2 tsz | tee -a $O/band1.txt
