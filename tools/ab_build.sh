#!/bin/bash
# Builds hmvec_amd/libhmgrid_base.so from the committed (HEAD) sources so that a working-tree change can be A/B-timed
# against it on the same GPU box in one call (box-to-box noise is +-4 %): tools/ab_run.sh alternates the two.
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
T=$(mktemp -d)
git -C $ROOT archive HEAD hmvec_amd/csrc include | tar -x -C $T
make -j4 -C $T/hmvec_amd/csrc OUT=$ROOT/hmvec_amd/libhmgrid_base.so 2>&1 | grep -E "error|warning" || true
rm -rf $T
ls -la $ROOT/hmvec_amd/libhmgrid_base.so
