#!/bin/bash
mkdir -p gpurun_out
python tools/chain_bench.py > gpurun_out/r3_chain_a.json 2>/dev/null; cat gpurun_out/r3_chain_a.json
HIP_FORCE_DEV_KERNARG=1 python tools/chain_bench.py > gpurun_out/r3_chain_b.json 2>/dev/null; cat gpurun_out/r3_chain_b.json
HIP_FORCE_DEV_KERNARG=0 python tools/chain_bench.py > gpurun_out/r3_chain_c.json 2>/dev/null; cat gpurun_out/r3_chain_c.json
HIP_FORCE_DEV_KERNARG=1 TVC_FOLD_LN=0 python tools/update_bench.py 2>/dev/null
HIP_FORCE_DEV_KERNARG=0 TVC_FOLD_LN=0 python tools/update_bench.py 2>/dev/null
