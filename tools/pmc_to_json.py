#!/usr/bin/env python3
"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE CSVs -> profiles/pmc_traffic.json (HBM bytes per launch).
gfx950 corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE is in KiB and reports exactly half of a
wide coalesced read stream -> x2; WRITE_SIZE (KiB) is exact for 16-byte-per-lane stores.
usage: pmc_to_json.py <fetch_dir> <write_dir> [out.json]"""
import csv, glob, json, os, re, sys
from collections import defaultdict


def collect(d, counter):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != counter:
                continue
            name = re.sub(r"\(anonymous namespace\)::|^void ", "", row["Kernel_Name"]).split("(")[0]
            if name.startswith("env_step_kernel<false") and name.rstrip().endswith("true>"):
                name = "env_step_kernel_w1000_dr"  # <W10 = false, DR = true>: the train loop's instantiation
            elif name.startswith("env_step_kernel<") and name.rstrip().endswith("true>"):
                name = "env_step_kernel_dr"  # the domain-randomised instantiation <W10, DR = true>
            elif "gemm_rowln" not in name:  # the fused kernel keeps its template arguments: <JT, k-tiles> tell the shapes apart
                name = name.split("<")[0]
            acc[(name, int(row["Grid_Size"]))].append(float(row["Counter_Value"]))
    return {k: sum(v[len(v) // 2:]) / len(v[len(v) // 2:]) for k, v in acc.items()}


sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tvc_ai_amd.build import sources_sha256  # noqa: E402
SHA = sources_sha256()  # bench.py prints "stale": true beside `traffic` when the library has moved on since this pass
fetch = collect(sys.argv[1], "FETCH_SIZE")
write = collect(sys.argv[2], "WRITE_SIZE")
out = defaultdict(dict)
for (name, grid), f in fetch.items():
    if (name, grid) not in write or not ("env_step" in name or "gemm_kernel" in name or "gemm_rowln" in name or "actor_rows" in name
                                         or "actor_split" in name or "actor_x3" in name):
        continue
    w = write[(name, grid)]
    out[name][str(grid)] = {"lib_sources_sha256": SHA, "fetch_size_kib_raw": f, "write_size_kib": w, "hbm_read_bytes": f * 1024 * 2,
                            "hbm_write_bytes": w * 1024, "hbm_bytes_per_launch": f * 1024 * 2 + w * 1024,
                            "hbm_bytes_per_thread": (f * 1024 * 2 + w * 1024) / grid}
path = sys.argv[3] if len(sys.argv) > 3 else "profiles/pmc_traffic.json"
if os.path.exists(path):  # merge into what earlier passes measured
    old = json.load(open(path))
    for k, v in old.items():
        for g, e in v.items():
            out[k].setdefault(g, e)
json.dump(out, open(path, "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))
