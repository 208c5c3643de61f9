#!/usr/bin/env python3
"""rocprofv3 --kernel-trace --stats of `bench.py --loop-only` -> profiles/r03_train_loop_kernel_stats.json:
{envs: {kernel name without template arguments / parameters: average microseconds in the train loop}}.
usage: loop_stats_to_json.py <kernel_stats.csv> <envs> [out.json]"""
import csv, json, os, re, sys
src, envs = sys.argv[1], sys.argv[2]
out = sys.argv[3] if len(sys.argv) > 3 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles",
                                                         "r03_train_loop_kernel_stats.json")
data = json.load(open(out)) if os.path.exists(out) else {}
rows = {}
for r in csv.DictReader(open(src)):
    name = r["Name"]
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)            # parameters
    short = re.sub(r"\(anonymous namespace\)::", "", name)
    rows[short] = float(r["AverageNs"]) / 1e3
    rows.setdefault(re.sub(r"<.*$", "", short), float(r["AverageNs"]) / 1e3)
# the acting pass is split into two launches per step since round 2 (some rows beside the learner, the rest alone): store the
# PER-STEP total under the plain kernel name so that it compares with one isolated launch over all rows.  Steps = launches of a
# kernel that runs exactly once per train step and never outside one (update_prep_kernel: one update per step; the env's own
# step kernel also runs in the prefill).  Profile with a FIXED split (--share-rows): the tuning at warm-up runs other splits.
calls = {r["Name"]: (int(r["Calls"]), float(r["TotalDurationNs"])) for r in csv.DictReader(open(src))}
steps = next((c for k, (c, _) in calls.items() if "update_prep_kernel" in k), None)
for k in ("tvcnn::actor_rows_kernel", "tvcnn::actor_x3_kernel"):
    full = next((n for n in calls if n.startswith(k) or n.startswith("void " + k)), None)  # (template instantiations carry the return type)
    if steps and full and calls[full][0] > steps:
        rows[k + " [average of one launch]"] = rows[k]
        rows[k] = calls[full][1] / 1e3 / steps
        rows.setdefault("_launches_per_step", {})[k] = calls[full][0] / steps
        rows["_train_steps_profiled"] = steps
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tvc_ai_amd.build import sources_sha256  # noqa: E402
rows["_lib_sources_sha256"] = sources_sha256()  # bench.py prints "stale": true beside in_loop_us when the library has moved on
data[str(envs)] = rows
json.dump(data, open(out, "w"), indent=1, sort_keys=True)
print("wrote", out, len(rows), "kernels")
