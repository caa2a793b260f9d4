#!/bin/bash
# The GPU suite under the library's testing switches (each reroutes part of the path; results must stay inside the same gates).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/envm
for e in ${ENVS:-"HMG_NO_GROUPS=1" "HMG_NO_HINTS=1" "HMG_FUSED_GENERIC=1" "HMG_CHIRP=0" "HMG_PRUNED_FFT=0" "HMG_BAND_FFT=0" "HMG_FUSED_FFT=0" "HMG_FORCE_GATHERV=1" "HMG_NO_ROWSC=1" "HMG_NO_TENSOR_GROUP=1"}; do
  env $e timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/envm/env_$e.log 2>&1
  echo "$e: $(tail -1 gpurun_out/envm/env_$e.log)"
done
