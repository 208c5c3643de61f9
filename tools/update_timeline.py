#!/usr/bin/env python3
"""Prints the kernel sequence of ONE SAC update from a rocprofv3 kernel-trace CSV of tools/learner_only.py:
start offset, duration, gap to the previous kernel, name, grid.  usage: update_timeline.py <kernel_trace.csv> [out.md]"""
import csv
import sys

from summarize_rocprof import short

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        g = "x".join(str(int(r[f"Grid_Size_{a}"]) // max(1, int(r[f"Workgroup_Size_{a}"]))) for a in "XYZ")
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), g))
rows.sort()
# an update starts with update_prep_kernel; take the last complete one
ends = [i - 1 for i, r in enumerate(rows) if "update_prep" in r[2]]
lo, hi = ends[-2] + 1, ends[-1] + 1
seq = rows[lo:hi]
t0 = seq[0][0]
out = ["| # | start us | dur us | gap us | kernel | grid |", "|---|---|---|---|---|---|"]
prev_end = t0
busy = 0
for i, (s, e, n, g) in enumerate(seq):
    out.append(f"| {i} | {(s-t0)/1e3:.1f} | {(e-s)/1e3:.2f} | {(s-prev_end)/1e3:.2f} | {n} | {g} |")
    busy += e - s
    prev_end = e
out.append(f"\n{len(seq)} kernels, span {(seq[-1][1]-t0)/1e3:.1f} us, kernel-busy {busy/1e3:.1f} us")
txt = "\n".join(out) + "\n"
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(txt)
print(txt)
