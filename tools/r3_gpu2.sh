#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
python tools/update_bench.py > gpurun_out/r3_upd_fold1.json 2> gpurun_out/r3_upd_fold1.err; echo "upd fold rc=$?"; cat gpurun_out/r3_upd_fold1.json
TVC_FOLD_LN=0 python tools/update_bench.py > gpurun_out/r3_upd_fold0.json 2> gpurun_out/r3_upd_fold0.err; echo "upd nofold rc=$?"; cat gpurun_out/r3_upd_fold0.json
python -m pytest tests/test_sac_parity_gpu.py tests/test_trainer_gpu.py tests/test_hier_parity_gpu.py tests/test_dp_gpu.py -x -q > gpurun_out/r3_t2_sac.log 2>&1; echo "sac tests rc=$?"
tail -12 gpurun_out/r3_t2_sac.log
python -m pytest tests/test_schedule_gpu.py -x -q > gpurun_out/r3_t2_schedule.log 2>&1; echo "schedule tests rc=$?"
tail -12 gpurun_out/r3_t2_schedule.log
