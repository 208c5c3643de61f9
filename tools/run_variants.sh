mkdir -p gpurun_out
echo "== calibration: per-layer kernels"
TVC_ROWS_MIN=1000000000 timeout -k 10 120 python tools/act_bench.py 65536 2>&1 | grep rows
for v in ${VARIANTS:-0 1 2 3 4}; do
  echo "== variant $v"
  TVC_ROWS_MIN=1 TVC_ROWS_VARIANT=$v timeout -k 10 120 python tools/act_bench.py 16384 65536 2>&1 | grep -v amdgpu.ids
done
echo "== calibration again"
TVC_ROWS_MIN=1000000000 timeout -k 10 120 python tools/act_bench.py 65536 2>&1 | grep rows
