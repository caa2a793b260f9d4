#!/bin/bash
# A/B on one box by environment switch: alternates `env A` and `env B` runs of the slab sweep.
# Usage: tools/env_ab.sh "HMG_NO_ROWSC=1" "" [rounds]      (an empty string = no extra variable)
set -e
A="$1"; B="$2"; R=${3:-2}
for i in $(seq $R); do
  for v in "$A" "$B"; do
    echo "== env [$v]"
    env $v SLABS="${SLABS:-32 4}" bash tools/slab_sweep.sh
  done
done
