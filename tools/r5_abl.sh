#!/bin/bash
# Timing ablations of the long-grid kernels (HMG_LG_ABL bits in longgrid.hip: 2 no residue-twiddle fetch, 4 no LDS passes,
# 8 no unpack / no band accumulation, 16 no mode loads in the interpolation, 32 no group loop, 64 no transcendentals).
# Build the variants first: for a in 4 8 12 44; do tools/long_build.sh abl$a "-DHMG_LG_ABL=$a"; done
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5; mkdir -p $O
echo "# HMG_CHIRP=0: every row takes the decomposition" | tee $O/abl.txt
HMG_CHIRP=0 bash tools/long_ab.sh "main abl4 abl8 abl12 abl44" 2 gas | tee -a $O/abl.txt
echo "# tSZ band kernel" | tee -a $O/abl.txt
bash tools/long_ab.sh "main abl4 abl8 abl64" 2 tsz | tee -a $O/abl.txt
