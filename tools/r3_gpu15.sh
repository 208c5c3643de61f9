#!/bin/bash
python -m pytest tests/test_env_parity_gpu.py tests/test_step_golden_gpu.py tests/test_env_semantics_gpu.py -q 2>&1 | tail -2
python bench.py --steps 300 --warmup 50 --loop-only --share-rows 32768 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('r3 65536 prefilled:', round(d['ms_per_step'],4))"
python bench.py --steps 300 --warmup 50 --loop-only 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('r3 65536 prefilled + tuned:', round(d['ms_per_step'],4), d['sac']['acting_rows_in_sharing_form'])"
python bench.py --steps 300 --warmup 50 --loop-only --prefill-steps 0 --share-rows 32768 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('r3 65536 no prefill:', round(d['ms_per_step'],4))"
(cd .r2ref && python bench.py --steps 300 --warmup 1200 --loop-only 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('r2 65536 warmup 1200:', round(d['ms_per_step'],4))")
python bench.py --envs-per-gpu 4096 --steps 300 --warmup 50 --loop-only --segments on 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('r3 4096 prefilled:', round(d['ms_per_step'],4))"
python bench.py --envs-per-gpu 8192 --steps 300 --warmup 50 --loop-only --segments on 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('r3 8192 prefilled:', round(d['ms_per_step'],4))"
python tools/env_window_time.py 2>/dev/null | tail -5
