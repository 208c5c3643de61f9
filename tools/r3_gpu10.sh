#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_schedule_gpu.py tests/test_sac_parity_gpu.py tests/test_hier_parity_gpu.py -q > gpurun_out/r3_t10.log 2>&1; echo "tests rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/r3_t10.log | tail
grep -E "^E " gpurun_out/r3_t10.log | head -10
python bench.py --steps 100 --warmup 30 --loop-only --acting-dropout > gpurun_out/r3_b10_65536_drop.json 2>gpurun_out/r3_b10_65536_drop.err
python bench.py --envs-per-gpu 4096 --steps 300 --warmup 40 --loop-only --segments on --acting-dropout > gpurun_out/r3_b10_4096_drop.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3_b10_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d["ms_per_step"],4), "ms/step", round(d["value"]/1e6,2), "M env-steps/s", "acting_dropout", d["sac"]["acting_dropout"], "share_rows", d["sac"]["acting_rows_in_sharing_form"])
    except Exception as e: print(f, "ERR", e)
PY
tail -3 gpurun_out/r3_b10_65536_drop.err
