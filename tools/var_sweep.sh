#!/bin/bash
# A/B/C on one box: the slab sweep for several builds of the library (HMG_LIB_PATH), alternating.
# Usage: tools/var_sweep.sh "a c" [rounds]   -> hmvec_amd/libhmgrid_<v>.so; "new" = the working-tree build
set -e
VARS=${1:-"new"}; R=${2:-2}
for i in $(seq $R); do
  for v in $VARS; do
    if [ $v = new ]; then unset HMG_LIB_PATH; else export HMG_LIB_PATH=$PWD/hmvec_amd/libhmgrid_$v.so; fi
    echo "== build $v"
    SLABS="${SLABS:-32 4}" bash tools/slab_sweep.sh
  done
done
