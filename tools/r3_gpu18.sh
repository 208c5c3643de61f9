#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/g18; rm -rf $O; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_sac_parity_gpu.py tests/test_hier_parity_gpu.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
TVC_LN_TAIL=0 python tools/update_bench.py 2>/dev/null | tee $O/upd_tail0.json
TVC_LN_TAIL=1 python tools/update_bench.py 2>/dev/null | tee $O/upd_tail1.json
TVC_LN_TAIL=0 python tools/update_bench.py 2>/dev/null | tee -a $O/upd_tail0.json
TVC_LN_TAIL=1 python tools/update_bench.py 2>/dev/null | tee -a $O/upd_tail1.json
timeout -k 10 500 python tools/cumask_shard.py 4096 8192 2>&1 | grep -v "^ *$" | tee $O/cumask.jsonl
for t in 0 4; do
  TVC_STEPS_IN_FLIGHT=$t timeout -k 10 400 python bench.py 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('inflight $t', d['ms_per_step'], json.dumps(d['shard_sizes']))" | tee -a $O/throttle.txt
done
