#!/bin/bash
python -m pytest tests/test_sac_parity_gpu.py tests/test_trainer_gpu.py tests/test_dp_gpu.py tests/test_schedule_gpu.py -q 2>&1 | tail -3
python tools/update_bench.py 2>/dev/null
TVC_UPDATE_AUX=0 python tools/update_bench.py 2>/dev/null
b() { python bench.py --envs-per-gpu $1 --steps 300 --warmup 50 --loop-only $3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$2 $1:', round(d['ms_per_step'],4))"; }
b 4096 aux "--segments on"; b 8192 aux "--segments on"; b 65536 aux ""
export TVC_UPDATE_AUX=0; b 4096 noaux "--segments on"; b 8192 noaux "--segments on"; b 65536 noaux ""
