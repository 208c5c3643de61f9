#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_dp_gpu.py -q > gpurun_out/r3_t12.log 2>&1; echo "dp tests rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/r3_t12.log | tail -5; grep -E "^E " gpurun_out/r3_t12.log | head
TVC_FORCE_DEVICE=0 TVC_DIST_BACKEND=gloo timeout -k 10 400 python bench.py --gpus 2 --envs-per-gpu 8192 --steps 100 --warmup 20 --loop-only > gpurun_out/r3_b12_2rank.json 2> gpurun_out/r3_b12_2rank.err; echo "2-rank rehearsal rc=$?"
tail -3 gpurun_out/r3_b12_2rank.err
python - <<'PY'
import json
try:
    d=json.loads(open("gpurun_out/r3_b12_2rank.json").read().strip().splitlines()[-1])
    print("2 ranks on one card (gloo):", d["n_gpus"], "ranks", round(d["ms_per_step"],4), "ms/step", d["config"]["workload"][-80:], d["allreduce"].get("measured_us_per_update"))
except Exception as e: print("ERR", e)
PY
