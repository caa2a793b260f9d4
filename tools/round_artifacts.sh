#!/bin/bash
# Judged artefacts of a round on the GPU box (run through gpurun): full GPU suite, smoke, the profile set of the headline
# step (tools/profile_round.sh: PMC traffic, SQ counters, bench line, kernel trace), slab sweep + thin-slab kernel
# times + thin-slab dispatch timeline, shape sweep, and kernel-stats / PMC / SQ summaries of the three long-grid
# launches (gas 30000/50, numeric NFW 40000/200, tSZ 30000/2).  Everything lands under gpurun_out/prof/: copy what is
# to be judged into profiles/$ROUND/.   Usage: ROUND=r06 bash tools/round_artifacts.sh
set -e
ROUND=${ROUND:-r06}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/final; mkdir -p $O gpurun_out/prof
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
python -c "import __graft_entry__ as g; g.smoke()" | tee $O/smoke.txt
ROUND=$ROUND bash tools/profile_round.sh > $O/profile_round.log 2>&1 || { tail -20 $O/profile_round.log; exit 1; }
tail -4 $O/profile_round.log
bash tools/slab_sweep.sh > gpurun_out/prof/slab_sweep.txt 2>&1; cat gpurun_out/prof/slab_sweep.txt
bash tools/kernel_times.sh "4" > gpurun_out/prof/slab4_kernel_times.txt 2>&1; cp gpurun_out/kt_4/k_kernel_stats.csv gpurun_out/prof/slab4_kernel_stats.csv; cat gpurun_out/prof/slab4_kernel_times.txt
python3 tools/trace_timeline.py gpurun_out/kt_4/k_kernel_trace.csv > gpurun_out/prof/slab4_timeline.txt; tail -12 gpurun_out/prof/slab4_timeline.txt
for w in gas nfw tsz; do
  pat=pruned_kernel; [ $w = tsz ] && pat=band_kernel
  bash tools/kernel_counters.sh $O/cnt_$w $pat tools/probes/stage_only.py $w 3 > $O/cnt_$w.txt 2>&1 || { tail -5 $O/cnt_$w.txt; exit 1; }
  cp $O/cnt_$w/summary.txt gpurun_out/prof/long_${w}_counters.txt; cp $O/cnt_$w/kernel_stats.csv gpurun_out/prof/long_${w}_kernel_stats.csv
done
python3 tools/shape_sweep.py > gpurun_out/prof/shape_sweep.txt 2> $O/shape_sweep.err || { tail -5 $O/shape_sweep.err; exit 1; }
tail -8 gpurun_out/prof/shape_sweep.txt
cp hmvec_amd/csrc/hmgrid.resources.txt gpurun_out/prof/ 2>/dev/null || true
echo "round artefacts done"
