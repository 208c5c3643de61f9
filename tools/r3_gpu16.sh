#!/bin/bash
b() { python bench.py --envs-per-gpu $1 --steps 300 --warmup 50 --loop-only --segments on 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$2 $1:', round(d['ms_per_step'],4), d['sac']['acting_rows_in_sharing_form'])"; }
b 4096 "split"; b 8192 "split"
export TVC_ROWS_MIN=4096; b 4096 "rows-kernel"; b 8192 "rows-kernel"; unset TVC_ROWS_MIN
python -m pytest tests/test_schedule_gpu.py -q -k "train_mode or vec_trainer" 2>&1 | tail -2
