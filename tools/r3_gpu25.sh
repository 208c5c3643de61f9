#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/g25; rm -rf $O; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_schedule_gpu.py tests/test_trainer_gpu.py tests/test_dp_gpu.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
timeout -k 10 500 python bench.py > $O/default_bench.json 2> $O/default_bench.err; python -c "
import json,sys
d=json.loads(open('$O/default_bench.json').read().strip().splitlines()[-1]); print('default bench', d['ms_per_step'], d['value'], json.dumps(d['sac']['streams']), json.dumps(d['sac']['cu_split']), json.dumps(d['shard_sizes']))" | tee -a $O/bench.txt
for n in 4096 8192; do python bench.py --loop-only --envs-per-gpu $n --steps 300 --warmup 30 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print($n, d['ms_per_step'], json.dumps(d['sac']['cu_split']))" | tee -a $O/bench.txt; done
