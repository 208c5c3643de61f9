#!/usr/bin/env python3
"""rocprofv3 --kernel-trace --stats of `bench.py --loop-only` -> profiles/r02_train_loop_kernel_stats.json:
{envs: {kernel name without template arguments / parameters: average microseconds in the train loop}}.
usage: loop_stats_to_json.py <kernel_stats.csv> <envs> [out.json]"""
import csv, json, os, re, sys
src, envs = sys.argv[1], sys.argv[2]
out = sys.argv[3] if len(sys.argv) > 3 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles",
                                                         "r02_train_loop_kernel_stats.json")
data = json.load(open(out)) if os.path.exists(out) else {}
rows = {}
for r in csv.DictReader(open(src)):
    name = r["Name"]
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)            # parameters
    short = re.sub(r"\(anonymous namespace\)::", "", name)
    rows[short] = float(r["AverageNs"]) / 1e3
    rows.setdefault(re.sub(r"<.*$", "", short), float(r["AverageNs"]) / 1e3)
data[str(envs)] = rows
json.dump(data, open(out, "w"), indent=1, sort_keys=True)
print("wrote", out, len(rows), "kernels")
