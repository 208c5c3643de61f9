#!/usr/bin/env python3
"""Where one train step spends its time, measured with HIP events on the two streams of VecTrainer (no profiler):
acting launch 1 (rows that share their CUs with the learner), acting launch 2 (whole chip), env step + replay insert, the SAC
update on the side stream, and the whole step.  usage (GPU box): python tools/step_events.py [envs=65536] [updates_per_step=1]"""
import json
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tvc_ai_amd.trainer import bench_train  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
utd = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = torch.device("cuda:0")
args = SimpleNamespace(family=0, envs_per_gpu=n, dr_stage=5, exact_reward=False, shipped_acting=False, updates_per_step=utd,
                       no_overlap=False, share_cus="auto")
w = bench_train(args, 1, 0, dev)
tr = w["trainer"]
for _ in range(30):
    tr.step(True)
torch.cuda.synchronize()
rec = []
cur = {}
ev = lambda: torch.cuda.Event(enable_timing=True)
act0, upd0, envstep0 = tr.sac.act, tr.sac.update, tr.env.step


def act(*a, **k):
    e0, e1 = ev(), ev()
    e0.record(); out = act0(*a, **k); e1.record()
    cur.setdefault("act", []).append((e0, e1))
    return out


def update(*a, **k):
    e0, e1 = ev(), ev()
    e0.record(); out = upd0(*a, **k); e1.record()   # recorded on the current (side) stream
    cur.setdefault("upd", []).append((e0, e1))
    return out


def envstep(*a, **k):
    e0 = ev(); e0.record(); out = envstep0(*a, **k)
    cur["env0"] = e0
    return out


tr.sac.act, tr.sac.update, tr.env.step = act, update, envstep
K = 100
for _ in range(K):
    cur = {}
    s0, s1 = ev(), ev()
    s0.record(); tr.step(True); s1.record()
    cur["step"] = (s0, s1)
    rec.append(cur)
torch.cuda.synchronize()
rows = []
for c in rec:
    s0, s1 = c["step"]
    r = {"step_us": s0.elapsed_time(s1) * 1e3}
    for i, (e0, e1) in enumerate(c["act"]):
        r[f"act{i}_start"] = s0.elapsed_time(e0) * 1e3
        r[f"act{i}_us"] = e0.elapsed_time(e1) * 1e3
    r["after_acting_us"] = c["act"][-1][1].elapsed_time(s1) * 1e3
    r["upd_start"] = s0.elapsed_time(c["upd"][0][0]) * 1e3
    r["upd_us"] = c["upd"][0][0].elapsed_time(c["upd"][-1][1]) * 1e3
    r["upd_end"] = s0.elapsed_time(c["upd"][-1][1]) * 1e3
    rows.append(r)
med = {k: float(np.median([r[k] for r in rows])) for k in rows[0]}
print(json.dumps({"envs": n, "updates_per_step": utd, "share_rows": tr.share_rows if tr.share_cus else 0, "median_us": med}, indent=1))
