#!/bin/bash
# full GPU suite, then the judged profile set of the round (tools/profile_round.sh), slab sweep and thin-slab kernel times
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4; mkdir -p $O gpurun_out/prof
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/g7_all.log 2>&1 || { tail -40 $O/g7_all.log; exit 1; }
tail -2 $O/g7_all.log
python -c "import __graft_entry__ as g; g.smoke()"
ROUND=r04 bash tools/profile_round.sh > $O/profile_round.log 2>&1 || { tail -20 $O/profile_round.log; exit 1; }
tail -4 $O/profile_round.log
bash tools/slab_sweep.sh > gpurun_out/prof/slab_sweep.txt 2>&1; cat gpurun_out/prof/slab_sweep.txt
bash tools/kernel_times.sh "4" > gpurun_out/prof/slab4_kernel_times.txt 2>&1; cp gpurun_out/kt_4/k_kernel_stats.csv gpurun_out/prof/slab4_kernel_stats.csv; cat gpurun_out/prof/slab4_kernel_times.txt
PMC_BENCH_FLAGS="--nxs 30000 --xmax 50" bash tools/pmc_kernel.sh pruned > gpurun_out/prof/sq_counters_pruned_nxs30000.txt 2>&1; cat gpurun_out/prof/sq_counters_pruned_nxs30000.txt
