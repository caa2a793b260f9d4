#!/bin/bash
# Round-5 judged artefacts on the GPU box: full GPU suite, smoke, the profile set of the headline step
# (tools/profile_round.sh -> profiles/r05 inputs), slab sweep + thin-slab kernel times, shape sweep, and kernel-stats /
# PMC / SQ summaries of the three long-grid launches (gas 30000/50, numeric NFW 40000/200, tSZ 30000/2).
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5f; mkdir -p $O gpurun_out/prof
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
python -c "import __graft_entry__ as g; g.smoke()" | tee $O/smoke.txt
ROUND=r05 bash tools/profile_round.sh > $O/profile_round.log 2>&1 || { tail -20 $O/profile_round.log; exit 1; }
tail -4 $O/profile_round.log
bash tools/slab_sweep.sh > gpurun_out/prof/slab_sweep.txt 2>&1; cat gpurun_out/prof/slab_sweep.txt
bash tools/kernel_times.sh "4" > gpurun_out/prof/slab4_kernel_times.txt 2>&1; cp gpurun_out/kt_4/k_kernel_stats.csv gpurun_out/prof/slab4_kernel_stats.csv; cat gpurun_out/prof/slab4_kernel_times.txt
for w in gas nfw tsz; do
  pat=pruned_kernel; [ $w = tsz ] && pat=band_kernel
  bash tools/kernel_counters.sh $O/cnt_$w $pat tools/probes/stage_only.py $w 3 > $O/cnt_$w.txt 2>&1 || { tail -5 $O/cnt_$w.txt; exit 1; }
  cp $O/cnt_$w/summary.txt gpurun_out/prof/long_${w}_counters.txt; cp $O/cnt_$w/kernel_stats.csv gpurun_out/prof/long_${w}_kernel_stats.csv
done
python3 tools/shape_sweep.py > gpurun_out/prof/shape_sweep.txt 2> $O/shape_sweep.err || { tail -5 $O/shape_sweep.err; exit 1; }
tail -8 gpurun_out/prof/shape_sweep.txt
echo "r5 final done"
