#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV per (kernel, grid size): calls, mean/median/min duration.
usage: summarize_rocprof.py <kernel_trace.csv> [out.md]"""
import csv
import re
import statistics as st
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z0-9_:<>, ]+?)\(", name)
    s = m.group(1) if m else name
    return s[:70]


def main():
    path = sys.argv[1]
    groups = defaultdict(list)
    meta = {}
    with open(path) as f:
        for row in csv.DictReader(f):
            if "Grid_Size_X" in row:
                g = "x".join(str(int(row[f"Grid_Size_{a}"]) // max(1, int(row[f"Workgroup_Size_{a}"]))) for a in "XYZ")
                wg = int(row["Workgroup_Size_X"])
            else:
                g, wg = row.get("Grid_Size", "?"), int(row.get("Workgroup_Size", 0) or 0)
            key = (short(row["Kernel_Name"]), g, wg)
            groups[key].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
            meta[key] = (row.get("VGPR_Count", "?"), row.get("SGPR_Count", "?"), row.get("LDS_Block_Size", "?"))
    lines = ["| kernel | grid (blocks x,y,z) | block | calls | mean us | median us | min us | VGPR | SGPR | LDS B |",
             "|---|---|---|---|---|---|---|---|---|---|"]
    for key, d in sorted(groups.items(), key=lambda kv: -sum(kv[1])):
        v = meta[key]
        lines.append(f"| {key[0]} | {key[1]} | {key[2]} | {len(d)} | {st.mean(d)/1e3:.2f} | {st.median(d)/1e3:.2f} | "
                     f"{min(d)/1e3:.2f} | {v[0]} | {v[1]} | {v[2]} |")
    out = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        with open(sys.argv[2], "a") as f:
            f.write(out)
    print(out)


if __name__ == "__main__":
    main()
