#!/bin/bash
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r5
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r5/readme_tests.log 2>&1 || { tail -40 gpurun_out/r5/readme_tests.log; exit 1; }
tail -2 gpurun_out/r5/readme_tests.log
python3 tools/probes/readme_profile.py > gpurun_out/r5/readme_profile3.txt 2>&1; head -32 gpurun_out/r5/readme_profile3.txt
