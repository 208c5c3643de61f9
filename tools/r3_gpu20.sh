#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/g20; rm -rf $O; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_sac_parity_gpu.py tests/test_hier_parity_gpu.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
TVC_LN_TAIL=0 python tools/update_bench.py 2>/dev/null | tee $O/upd_tail0.json
TVC_LN_TAIL=1 python tools/update_bench.py 2>/dev/null | tee $O/upd_tail1.json
TVC_LN_TAIL=0 python tools/update_bench.py 2>/dev/null | tee -a $O/upd_tail0.json
TVC_LN_TAIL=1 python tools/update_bench.py 2>/dev/null | tee -a $O/upd_tail1.json
for lib in rb24 rb32 rb48; do
  export TVC_HIP_LIB=$PWD/tvc_ai_amd/csrc/libtvc_hip_$lib.so
  echo "lib=$lib" | tee -a $O/ring.txt; timeout -k 10 200 python tools/env_ring_bench.py 4096 65536 1048576 2>&1 | grep -v "^ *$" | tee -a $O/ring.txt
done
unset TVC_HIP_LIB
for c in "0 0" "4 0" "8 0" "16 0" "0 4" "0 8" "8 8" "30 30"; do timeout -k 10 100 python tools/stream_count.py $c 2>&1 | grep "streams touched" | tee -a $O/streams.txt; done
for c in "8 8" "30 30"; do GPU_MAX_HW_QUEUES=8 timeout -k 10 100 python tools/stream_count.py $c 2>&1 | grep "streams touched" | tee -a $O/streams.txt; done
