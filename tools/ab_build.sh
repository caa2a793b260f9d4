#!/bin/bash
# Builds hmvec_amd/libhmgrid_base.so from the committed (HEAD) kernel sources so that a working-tree
# change can be A/B-timed against it on the same GPU box in one call (box-to-box noise is +-4 %).
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
T=$(mktemp -d)
mkdir -p $T/hmvec_amd/csrc $T/include
for f in hmgrid.hip longgrid.hip longgrid.hpp rowdev.hpp sici.hpp ldsfft.hpp fastmath.hpp Makefile; do git -C $ROOT show HEAD:hmvec_amd/csrc/$f > $T/hmvec_amd/csrc/$f; done
git -C $ROOT show HEAD:include/hmgrid.h > $T/include/hmgrid.h
make -j2 -C $T/hmvec_amd/csrc OUT=$ROOT/hmvec_amd/libhmgrid_base.so 2>&1 | grep -E "error|warning" || true
rm -rf $T
ls -la $ROOT/hmvec_amd/libhmgrid_base.so
