#!/usr/bin/env python3
"""Prints the kernels of ONE train step (between two consecutive env_step launches of the main loop) from a
rocprofv3 kernel-trace CSV of bench.py, per HIP queue: start offset, duration, name, grid.
usage: step_timeline.py <kernel_trace.csv> <env-blocks> [out.md]"""
import csv
import sys
from collections import Counter

from summarize_rocprof import short

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        g = "x".join(str(int(r[f"Grid_Size_{a}"]) // max(1, int(r[f"Workgroup_Size_{a}"]))) for a in "XYZ")
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), g, r.get("Queue_Id", "?")))
rows.sort()
blocks = sys.argv[2] + "x1x1"
marks = [i for i, r in enumerate(rows) if "env_step_kernel" in r[2] and r[3] == blocks]
# a train step (many kernels between two env launches) in the middle of the timed region
wide = [j for j in range(len(marks) - 1) if marks[j + 1] - marks[j] > 50]
k = wide[len(wide) // 2]
lo, hi = marks[k] + 1, marks[k + 1] + 1
seq = rows[lo:hi]
t0 = rows[marks[k]][1]
out = [f"step window = {(seq[-1][1]-t0)/1e3:.1f} us, {len(seq)} kernels", "",
       "| start us | dur us | queue | kernel | grid |", "|---|---|---|---|---|"]
busy = Counter()
for s, e, n, g, q in seq:
    out.append(f"| {(s-t0)/1e3:.1f} | {(e-s)/1e3:.2f} | {q} | {n} | {g} |")
    busy[q] += e - s
out.append("")
for q, b in busy.items():
    out.append(f"queue {q}: kernel-busy {b/1e3:.1f} us")
txt = "\n".join(out) + "\n"
if len(sys.argv) > 3:
    open(sys.argv[3], "w").write(txt)
print(txt)
