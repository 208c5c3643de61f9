#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/g23; rm -rf $O; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
GPU_MAX_HW_QUEUES=4 timeout -k 10 100 python tools/stream_count.py 4 0 2>&1 | grep "streams touched" | tee -a $O/streams.txt
GPU_MAX_HW_QUEUES=4 timeout -k 10 100 python tools/stream_count.py 0 4 2>&1 | grep "streams touched" | tee -a $O/streams.txt
timeout -k 10 100 python tools/stream_count.py 4 0 2>&1 | grep "streams touched" | tee -a $O/streams.txt
timeout -k 10 400 python bench.py > $O/default_bench.json 2> $O/default_bench.err; python -c "
import json,sys
d=json.loads(open('$O/default_bench.json').read().strip().splitlines()[-1]); print('default bench', d['ms_per_step'], d['value'], json.dumps(d['sac']['streams']), json.dumps(d['shard_sizes']))" | tee -a $O/bench.txt
