#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/g24; rm -rf $O; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_sac_parity_gpu.py tests/test_hier_parity_gpu.py tests/test_trainer_gpu.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for v in 0 1 0 1; do TVC_THIN_LN=$v python tools/update_bench.py 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('thin_ln=$v', round(d['us_per_update'],1))" | tee -a $O/upd.txt; done
for n in 4096 8192; do
 for seg in 0 1; do
  for c in "-1 -1 -1 -1" "0 128 128 256" "0 96 96 256" "0 64 64 256" "0 160 160 256" "0 192 192 256" "0 128 -1 -1" "-1 -1 128 256"; do
    timeout -k 10 100 python tools/cumask_shard.py $n $seg $c 2>&1 | grep '"n"' | tee -a $O/cumask.jsonl
  done
 done
done
