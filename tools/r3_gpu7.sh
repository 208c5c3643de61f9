#!/bin/bash
mkdir -p gpurun_out
for n in 4096 8192; do
  for mode in "" "--segments on"; do
    tag=$(echo $mode | tr -d ' -')
    python bench.py --envs-per-gpu $n --steps 300 --warmup 40 --loop-only $mode > gpurun_out/r3_b_${n}_$tag.json 2> gpurun_out/r3_b_${n}_$tag.err; echo "bench $n $mode rc=$?"
  done
done
TVC_FOLD_LN=0 python bench.py --envs-per-gpu 4096 --steps 300 --warmup 40 --loop-only --segments on > gpurun_out/r3_b_4096_nofold.json 2>/dev/null
TVC_FOLD_LN=0 python bench.py --envs-per-gpu 8192 --steps 300 --warmup 40 --loop-only --segments on > gpurun_out/r3_b_8192_nofold.json 2>/dev/null
python bench.py --steps 100 --warmup 30 --loop-only > gpurun_out/r3_b_65536.json 2> gpurun_out/r3_b_65536.err; echo "bench 65536 rc=$?"
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3_b_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d["ms_per_step"],4), "ms/step", round(d["sac_updates_per_s"],1), "upd/s", "share_rows", d["sac"]["acting_rows_in_sharing_form"], d["config"]["workload"][-50:])
        t=d["sac"].get("share_rows_tuning")
        if t: print("   tuning:", [(c["share_rows"], round(c["us_per_step"])) for c in t["candidates"]], "slack", round(t["update_end_slack_us"]))
    except Exception as e: print(f, "ERR", e)
PY
python -m pytest tests/test_schedule_gpu.py -q > gpurun_out/r3_t7.log 2>&1; echo "schedule rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/r3_t7.log | tail
