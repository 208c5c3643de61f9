#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_sac_parity_gpu.py tests/test_trainer_gpu.py tests/test_dp_gpu.py -q > gpurun_out/r3_t11.log 2>&1; echo "tests rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/r3_t11.log | tail -5
python tools/update_bench.py 2>/dev/null
TVC_SKINNY_TILE=32 python tools/update_bench.py 2>/dev/null
python tools/update_bench.py 1 2>/dev/null
TVC_SKINNY_TILE=32 python tools/update_bench.py 1 2>/dev/null
python bench.py --envs-per-gpu 4096 --steps 300 --warmup 40 --loop-only --segments on 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('4096:', round(d['ms_per_step'],4))"
TVC_SKINNY_TILE=32 python bench.py --envs-per-gpu 4096 --steps 300 --warmup 40 --loop-only --segments on 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('4096 tile32:', round(d['ms_per_step'],4))"
