#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/g22; rm -rf $O; mkdir -p $O
run() { timeout -k 10 100 python tools/stream_count.py $3 $4 2>&1 | grep "streams touched" | sed "s/^/[Q=$1 PRIO=$2] /" | tee -a $O/streams.txt; }
for q in default 8 16; do
  for pr in -1 0; do
    for c in "0 0" "4 0" "0 4" "2 2" "1 0" "0 1"; do
      if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
      export TVC_SIDE_PRIORITY=$pr
      run $q $pr $c
    done
  done
done
