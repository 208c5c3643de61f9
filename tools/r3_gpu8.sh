#!/bin/bash
mkdir -p gpurun_out
echo "== default lib"; python tools/step_events.py 65536 2>/dev/null | tail -16
python bench.py --steps 100 --warmup 30 --loop-only --share-rows 32768 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('65536 share 32768:', round(d['ms_per_step'],4))"
echo "== lb4 lib"; export TVC_HIP_LIB=$PWD/tvc_ai_amd/csrc/libtvc_hip_lb4.so
python bench.py --steps 100 --warmup 30 --loop-only --share-rows 32768 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('65536 share 32768:', round(d['ms_per_step'],4))"
python bench.py --envs-per-gpu 4096 --steps 300 --warmup 40 --loop-only --segments on 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('4096:', round(d['ms_per_step'],4))"
python tools/update_bench.py 2>/dev/null
