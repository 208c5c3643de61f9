#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/g19; rm -rf $O; mkdir -p $O
export TVC_LN_TAIL=0
for v in base nocapture gc prio0 noreplay other_n eager_first; do timeout -k 10 120 python tools/anomaly2.py $v 2>&1 | grep -v "^ *$" | tail -4 | tee -a $O/anomaly.txt; done
for lib in rb16 "" rb128; do
  if [ -n "$lib" ]; then export TVC_HIP_LIB=$PWD/tvc_ai_amd/csrc/libtvc_hip_$lib.so; else unset TVC_HIP_LIB; fi
  echo "lib=$lib" | tee -a $O/ring.txt; timeout -k 10 200 python tools/env_ring_bench.py 4096 65536 1048576 2>&1 | grep -v "^ *$" | tee -a $O/ring.txt
done
unset TVC_HIP_LIB
PYTHONPATH=$PWD timeout -k 10 500 python tools/cumask_shard.py 4096 8192 2>&1 | grep -v "^ *$" | tee $O/cumask.jsonl
