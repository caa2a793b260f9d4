#!/usr/bin/env python3
"""Dispatch timeline of the bench step from a rocprofv3 --kernel-trace CSV (tools/kernel_times.sh leaves one under
gpurun_out/kt_<nz>/): finds the longest run of back-to-back steps (a step = the kernels from one front launch to the next,
no copy kernel in between: the timed graph-replay loop), prints three consecutive steps of it - start offset, duration
and the GAP to the previous kernel's end - and the averages over the whole run: per-kernel duration, per-boundary gap,
step period.  The gaps are what a launch boundary costs on top of the kernels' own ramp and drain."""
import csv
import sys
from collections import defaultdict

rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Kind"] == "KERNEL_DISPATCH"]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
short = lambda n: n.replace("void ", "").replace("hmg::", "").split("(")[0][:44]          # noqa: E731
K = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in rows]
starts = [i for i, k in enumerate(K) if k[2].startswith("front_group_kernel")]
steps = []                                   # (first index, one-past-last index) of steps made of hmg kernels only
for a, b in zip(starts, starts[1:]):
    if all("copyBuffer" not in K[i][2] and "fillBuffer" not in K[i][2] for i in range(a, b)):
        steps.append((a, b))
runs, cur = [], []
for s in steps:                              # runs of steps that follow each other directly with the same kernel count
    if cur and cur[-1][1] == s[0] and (s[1] - s[0]) == (cur[-1][1] - cur[-1][0]):
        cur.append(s)
    else:
        if cur:
            runs.append(cur)
        cur = [s]
if cur:
    runs.append(cur)
if not runs:
    sys.exit("no back-to-back steps found in the trace")
run = max(runs, key=len)
run = run[2:] if len(run) > 6 else run       # (the first steps of a run still see the previous phase's tail)
print(f"# {sys.argv[1]}: {len(run)} back-to-back steps of {run[0][1] - run[0][0]} launches each")
mid = len(run) // 2
t0 = K[run[mid][0]][0]
print("# three consecutive steps (microseconds from the first launch):   start      end      dur      gap")
for a, b in run[mid:mid + 3]:
    for i in range(a, b):
        gap = (K[i][0] - K[i - 1][1]) / 1e3
        print(f"  {K[i][2]:46s} {(K[i][0]-t0)/1e3:9.1f} {(K[i][1]-t0)/1e3:9.1f} {(K[i][1]-K[i][0])/1e3:8.1f} {gap:8.1f}")
    print()
dur, gaps = defaultdict(list), defaultdict(list)
for a, b in run:
    for i in range(a, b):
        dur[(i - a, K[i][2])].append((K[i][1] - K[i][0]) / 1e3)
        gaps[(i - a, K[i][2])].append((K[i][0] - K[i - 1][1]) / 1e3)
period = [(K[b][0] - K[a][0]) / 1e3 for a, b in run if b < len(K)]
import statistics as st
print("# medians over the run (under the profiler the host can fall behind a 0.1 ms step: the gap in front of a step's first")
print("# launch is then the host's, which is why means are not quoted):  duration   gap before")
tot_d = tot_g = 0.0
for key in sorted(dur):
    d, g = st.median(dur[key]), st.median(gaps[key])
    tot_d += d; tot_g += g
    print(f"  {key[1]:46s}           {d:8.1f}   {g:8.1f}")
print(f"  {'sum':46s}           {tot_d:8.1f}   {tot_g:8.1f}     median step period {st.median(period):.1f} us")
