#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/g21; rm -rf $O; mkdir -p $O
for lib in "" rb32 rb64; do
  if [ -n "$lib" ]; then export TVC_HIP_LIB=$PWD/tvc_ai_amd/csrc/libtvc_hip_$lib.so; else unset TVC_HIP_LIB; fi
  echo "lib=$lib" | tee -a $O/ring.txt; timeout -k 10 200 python tools/env_ring_bench.py 4096 65536 1048576 2>&1 | grep -v "^ *$" | grep -v amdgpu.ids | tee -a $O/ring.txt
done
unset TVC_HIP_LIB
timeout -k 10 300 python -m pytest tests/test_env_parity_gpu.py -x -q -m gpu > $O/pytest_env.log 2>&1; echo "pytest env rc=$?"; tail -3 $O/pytest_env.log
for c in "0 0" "4 0" "8 0" "0 4" "0 8" "3 5"; do timeout -k 10 100 python tools/stream_count.py $c 2>&1 | grep "streams touched" | tee -a $O/streams.txt; done
TVC_STREAM_PROBE=0 timeout -k 10 100 python tools/stream_count.py 4 0 2>&1 | grep "streams touched" | tee -a $O/streams.txt
timeout -k 10 400 python bench.py 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default bench', d['ms_per_step'], d['value'], json.dumps(d['shard_sizes']))" | tee -a $O/bench.txt
