#!/usr/bin/env python3
"""Print the kernel timeline of the last bench step from a rocprofv3 --kernel-trace CSV:
start offset, duration, stream/queue and name - to see which launches overlap."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
nlast = int(sys.argv[2]) if len(sys.argv) > 2 else 16
tail = rows[-nlast:]
t0 = int(tail[0]["Start_Timestamp"])
for r in tail:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{s/1e3:9.1f} {e/1e3:9.1f} us  dur {(e-s)/1e3:7.1f}  q{r.get('Queue_Id','?'):>3} s{r.get('Stream_Id','?'):>3}  {r['Kernel_Name'][:60]}")
