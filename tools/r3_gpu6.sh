#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_sac_parity_gpu.py -q -k "act_ or acting or forward" > gpurun_out/r3_t6.log 2>&1; echo "act tests rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/r3_t6.log | tail
echo "--- split kernel (default thresholds)"; python tools/act_bench.py 1024 2048 4096 6144 8192 2>/dev/null | grep rows
echo "--- split kernel forced at all sizes"; TVC_ROWS_MIN=1000000000 python tools/act_bench.py 12288 16384 32768 65536 2>/dev/null | grep rows
echo "--- rows kernel"; python tools/act_bench.py 12288 16384 32768 65536 2>/dev/null | grep rows
echo "--- per-layer kernels"; TVC_SPLIT_MIN=1000000000 python tools/act_bench.py 1024 4096 8192 2>/dev/null | grep rows
echo "--- split, share form"; TVC_ACT_SHARE=1 python tools/act_bench.py 4096 8192 2>/dev/null | grep rows
python tools/update_bench.py 2>/dev/null
TVC_FOLD_LN=0 python tools/update_bench.py 2>/dev/null
