#!/bin/bash
# Long-grid stage times of named build variants (hmvec_amd/libhmgrid_<v>.so; "main" = the working-tree build), alternating
# on one box.  Usage: tools/long_ab.sh "main base nolpt" [rounds] [which=gas,nfw,tsz]
VARS="$1"; R=${2:-2}; W=${3:-gas,nfw,tsz}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for i in $(seq $R); do
for v in $VARS; do
  if [ $v = main ]; then unset HMG_LIB_PATH; else export HMG_LIB_PATH=$PWD/hmvec_amd/libhmgrid_$v.so; fi
  timeout -k 10 300 python3 tools/probes/long_stage.py $v $W 2>/tmp/ls_$v.err || { echo "$v FAILED"; tail -5 /tmp/ls_$v.err; }
done; done
