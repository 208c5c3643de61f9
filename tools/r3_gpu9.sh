#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -m gpu -q > gpurun_out/r3_t9.log 2>&1; echo "gpu suite rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/r3_t9.log | tail
python bench.py --steps 200 --warmup 30 --loop-only > gpurun_out/r3_b9_65536.json 2>/dev/null
python bench.py --envs-per-gpu 4096 --steps 300 --warmup 40 --loop-only --segments on > gpurun_out/r3_b9_4096.json 2>/dev/null
python bench.py --envs-per-gpu 8192 --steps 300 --warmup 40 --loop-only --segments on > gpurun_out/r3_b9_8192.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3_b9_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d["ms_per_step"],4), "ms/step", round(d["sac_updates_per_s"],1), "upd/s", "share_rows", d["sac"]["acting_rows_in_sharing_form"])
        t=d["sac"].get("share_rows_tuning")
        if t: print("   tuning:", [(c["share_rows"], round(c["us_per_step"])) for c in t["candidates"]], "slack", round(t["update_end_slack_us"]))
    except Exception as e: print(f, "ERR", e)
PY
