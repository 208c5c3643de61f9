#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/g24; rm -rf $O; mkdir -p $O
for n in 4096 8192; do
 for seg in 0 1; do
  for c in "-1 -1 -1 -1" "0 128 128 256" "0 96 96 256" "0 64 64 256" "0 160 160 256" "0 192 192 256" "0 128 -1 -1" "-1 -1 128 256"; do
    timeout -k 10 100 python tools/cumask_shard.py $n $seg $c 2>&1 | grep '"n"' | tee -a $O/cumask.jsonl
  done
 done
done
