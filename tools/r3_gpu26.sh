#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/g26; rm -rf $O; mkdir -p $O
show() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], json.dumps(d['sac']['cu_split']), json.dumps(d['sac']['streams']['tuning']))" | tee -a $O/bench.txt; }
TVC_TUNE_STREAM=0 python bench.py --loop-only --envs-per-gpu 4096 --steps 300 --warmup 30 2>/dev/null | show "4096 no-stream-tune auto"
TVC_TUNE_STREAM=0 python bench.py --loop-only --envs-per-gpu 4096 --steps 300 --warmup 30 --cu-split 96 2>/dev/null | show "4096 no-stream-tune 96"
python bench.py --loop-only --envs-per-gpu 4096 --steps 300 --warmup 30 --cu-split 96 2>/dev/null | show "4096 stream-tune 96"
TVC_TUNE_STREAM=0 TVC_SIDE_PRIORITY=0 python bench.py --loop-only --envs-per-gpu 4096 --steps 300 --warmup 30 2>/dev/null | show "4096 no-stream-tune prio0 auto"
TVC_TUNE_STREAM=0 python bench.py --loop-only --envs-per-gpu 8192 --steps 300 --warmup 30 --cu-split 128 2>/dev/null | show "8192 no-stream-tune 128"
TVC_TUNE_STREAM=0 GPU_MAX_HW_QUEUES=4 python bench.py --loop-only --envs-per-gpu 4096 --steps 300 --warmup 30 --cu-split 96 2>/dev/null | show "4096 no-stream-tune Q4 96"
