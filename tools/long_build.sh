#!/bin/bash
# Variant builds of the library for A/B timing of the long-grid unit on one box:
#   tools/long_build.sh <name> "<LONG_EXTRA flags>" ["<LONG_LICM flags>"] -> hmvec_amd/libhmgrid_<name>.so
# (the headline unit hmgrid.o is rebuilt per variant too: objects are named after the output)
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
N=$1; X=$2
if [ $# -ge 3 ]; then
  make -j2 -C $ROOT/hmvec_amd/csrc OUT=../libhmgrid_$N.so LONG_EXTRA="$X" LONG_LICM="$3" 2>&1 | grep -E "error|warning: v|spill" || true
else
  make -j2 -C $ROOT/hmvec_amd/csrc OUT=../libhmgrid_$N.so LONG_EXTRA="$X" 2>&1 | grep -E "error" || true
fi
ls -la $ROOT/hmvec_amd/libhmgrid_$N.so
