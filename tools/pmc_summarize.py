#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs: mean counter value per (kernel, grid) over dispatches.
usage: pmc_summarize.py <dir-with-csvs> [kernel-name-substring, default env_step]"""
import csv, glob, os, re, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
want = sys.argv[2] if len(sys.argv) > 2 and not os.path.isdir(sys.argv[2]) else "env_step"
for d in [a for a in sys.argv[1:] if os.path.isdir(a)]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            name = re.sub(r"\(anonymous namespace\)::|^void ", "", row["Kernel_Name"]).split("(")[0][:48]
            key = (name, int(row["Grid_Size"]))
            acc[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
for key, ctrs in sorted(acc.items()):
    if want not in key[0]:
        continue
    print(key[0], "grid", key[1])
    for c, v in sorted(ctrs.items()):
        v = v[len(v) // 2:]  # second half of the dispatches (steady state)
        print(f"   {c:24s} mean {sum(v)/len(v):16.1f}   per-env {sum(v)/len(v)/key[1]:12.4f}  (n={len(v)})")
