#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_sac_parity_gpu.py tests/test_trainer_gpu.py tests/test_hier_parity_gpu.py tests/test_dp_gpu.py tests/test_schedule_gpu.py tests/test_config_sizes_gpu.py tests/test_aux_parity_gpu.py -q > gpurun_out/r3_t3.log 2>&1; echo "tests rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/r3_t3.log | tail -30
cd /tmp && export TMPDIR=/tmp
for f in 1 0; do
  rm -rf /tmp/prof_$f
  TVC_FOLD_LN=$f rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$f -- python3 $GRAFT_REPO_ROOT/tools/learner_only.py 30 0 0.1 > $GRAFT_REPO_ROOT/gpurun_out/r3_prof_$f.log 2>&1
  csv=$(find /tmp/prof_$f -name "*kernel_trace.csv" | head -1)
  (cd $GRAFT_REPO_ROOT/tools && python3 update_timeline.py $csv $GRAFT_REPO_ROOT/gpurun_out/r3_update_timeline_fold$f.md > /dev/null 2>&1); echo "timeline fold=$f rc=$?"
  tail -2 $GRAFT_REPO_ROOT/gpurun_out/r3_update_timeline_fold$f.md
done
