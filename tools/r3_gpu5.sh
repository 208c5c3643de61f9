#!/bin/bash
mkdir -p gpurun_out
python tools/chain_bench.py > gpurun_out/r3_chain_d.json 2>/dev/null; cat gpurun_out/r3_chain_d.json
python tools/gemm_stamps.py 512 256 256 2>/dev/null | grep -A6 median_phase
python tools/update_bench.py 2>/dev/null
TVC_FOLD_LN=0 python tools/update_bench.py 2>/dev/null
python -m pytest tests/test_sac_parity_gpu.py tests/test_trainer_gpu.py tests/test_hier_parity_gpu.py tests/test_dp_gpu.py -q > gpurun_out/r3_t5.log 2>&1; echo "tests rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/r3_t5.log | tail
