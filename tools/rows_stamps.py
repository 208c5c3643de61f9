#!/usr/bin/env python3
"""Where the one-launch acting kernel loses time at the launch level: per-workgroup start / end stamps (s_memrealtime, 100 MHz)
and hardware ids of the last of several back-to-back launches (tvc_debug_rows_stamps).
usage (GPU box): python tools/rows_stamps.py [rows=65536] [share=0]"""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tvc_ai_amd.agent import NativeSAC, sac_cfg  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
share = int(sys.argv[2]) if len(sys.argv) > 2 else 0
sac = NativeSAC(sac_cfg(0, batch_size=256, max_act_rows=n), device="cuda:0", seed=2)
ob = torch.randn(n, 10, device="cuda:0")
nwg = (n + 63) // 64
STRIDE = int(os.environ.get("TVC_STAMP_STRIDE", "6"))  # 96 with a library built with -DAR_TRACE (per-pass stamps of wave 0)
buf = (C.c_uint64 * (nwg * STRIDE))()
rc = sac.L.tvc_debug_rows_stamps(sac._h, ob.data_ptr(), n, 10, 4 if share else 0, buf, torch.cuda.current_stream().cuda_stream)
assert rc == 0, sac.L.tvc_last_error()
v = np.frombuffer(buf, dtype=np.uint64).reshape(nwg, STRIDE).astype(np.int64)
t0 = v[:, 1].min()
start, end = (v[:, 1] - t0) / 100.0, (v[:, 3] - t0) / 100.0   # microseconds
life = end - start
xcc, hw = v[:, 4] & 0xF, v[:, 5]
cu = (hw >> 8) & 0xF
se = (hw >> 13) & 0x7
sh = (hw >> 12) & 0x1
cuid = xcc * 1000 + se * 100 + sh * 10 + cu   # a label per physical CU
order = np.argsort(start)
first = start < np.percentile(life, 5) * 0.5    # dispatched before anything could have retired
rep = {"rows": n, "workgroups": nwg, "share_cus": bool(share), "span_us": float(end.max()),
       "first_wave": {"count": int(first.sum()), "start_us_p50": float(np.median(start[first])), "start_us_max": float(start[first].max()),
                      "life_us_p50": float(np.median(life[first])), "life_us_p5": float(np.percentile(life[first], 5)),
                      "life_us_p95": float(np.percentile(life[first], 95)), "end_us_max": float(end[first].max())},
       "later": ({"count": int((~first).sum()), "start_us_p5": float(np.percentile(start[~first], 5)),
                  "start_us_p50": float(np.median(start[~first])), "start_us_p95": float(np.percentile(start[~first], 95)),
                  "life_us_p50": float(np.median(life[~first])), "life_us_p5": float(np.percentile(life[~first], 5)),
                  "life_us_p95": float(np.percentile(life[~first], 95)), "end_us_p5": float(np.percentile(end[~first], 5)),
                  "end_us_p50": float(np.median(end[~first]))} if (~first).any() else None),
       "distinct_cus": int(len(np.unique(cuid))), "distinct_xcc": int(len(np.unique(xcc)))}
# per CU: busy time and the time it sat idle before the kernel ended
per = {}
for c in np.unique(cuid):
    m = cuid == c
    per[int(c)] = (int(m.sum()), float(end[m].max()), float(life[m].sum()))
wg_per_cu = np.array([p[0] for p in per.values()])
last_end = np.array([p[1] for p in per.values()])
rep["per_cu"] = {"workgroups_min": int(wg_per_cu.min()), "workgroups_max": int(wg_per_cu.max()),
                 "workgroups_hist": {int(k): int((wg_per_cu == k).sum()) for k in np.unique(wg_per_cu)},
                 "last_end_us_p5": float(np.percentile(last_end, 5)), "last_end_us_p50": float(np.median(last_end)),
                 "last_end_us_max": float(last_end.max())}
# concurrency over time: how many workgroups are alive, sampled every 10 us
ts = np.arange(0, end.max(), 10.0)
alive = np.array([((start <= t) & (end > t)).sum() for t in ts])
rep["alive_workgroups"] = {"mean": float(alive.mean()), "p50": float(np.median(alive)), "max": int(alive.max()),
                           "time_below_90pct_of_max_us": float(10.0 * (alive < 0.9 * alive.max()).sum())}
# which block ids share a CU in the first wave of dispatches (for de-phasing experiments)
pairs = {}
for b in np.where(first)[0]:
    pairs.setdefault(int(cuid[b]), []).append(int(b))
diffs = {}
for c, bs in pairs.items():
    if len(bs) == 2:
        d = abs(bs[1] - bs[0])
        diffs[d] = diffs.get(d, 0) + 1
rep["first_wave_block_id_distance_of_cu_mates"] = {int(k): int(v) for k, v in sorted(diffs.items(), key=lambda kv: -kv[1])[:6]}
rep["first_wave_examples"] = [pairs[c] for c in list(pairs)[:6]]
rep["xcc_of_block_0_to_15"] = [int(x) for x in xcc[:16]]
rep["mfma_cycles_floor_us_per_workgroup_alone"] = 402 * 64 * 32 / 2400.0
if STRIDE > 6:
    labels = ["thin pass (2 tiles)", "norm1"]
    for l in range(4):
        if l > 0:
            labels += [f"L{l} attention pass (16)", f"L{l} +res, norm1"]
        labels += [f"L{l} W1 half0 pass (16)", f"L{l} bias+GELU", f"L{l} W2 half0 pass (16)", f"L{l} W1 half1 pass (16)", f"L{l} bias+GELU",
                   f"L{l} W2 half1 pass (16)", f"L{l} +res, norm2"]
    labels += ["feature_norm", "head.0 pass a (16)", "head.0 pass b (16)", "bias+GELU+LayerNorm(512)", "head.4 half0 pass (32)",
               "GELU + running sums, head.4 half1 pass (32)"]
    t = v[:, 6:6 + len(labels)]
    prev = np.concatenate([v[:, 0:1], t[:, :-1]], axis=1)
    d = (t - prev).astype(np.float64)
    ok = (t > 0).all(axis=1)
    med = np.median(d[ok], axis=0)
    tiles = [int(x.split("(")[-1].rstrip(")")) if x.endswith(")") and x.split("(")[-1].rstrip(")").isdigit() else 0 for x in labels]
    rep["trace_cycles_median"] = [{"segment": lb, "cycles": float(m), "cycles_per_tile": (float(m) / tl if tl else None)} for lb, m, tl in zip(labels, med, tiles)]
    tot_pass = sum(m for m, tl in zip(med, tiles) if tl)
    tot_tiles = sum(tiles)
    rep["trace_summary"] = {"pass_cycles": float(tot_pass), "tiles": tot_tiles, "cycles_per_tile_in_passes": float(tot_pass / tot_tiles),
                            "epilogue_cycles": float(med.sum() - tot_pass), "total": float(med.sum())}
print(json.dumps(rep, indent=1))
sac.close()
