# A/B of acting-kernel variants: prebuilt libraries (LIBS, via TVC_HIP_LIB) x start-stagger settings (STAGGERS, via TVC_ROWS_STAGGER)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for lib in ${LIBS:-libtvc_hip.so}; do
  for sg in ${STAGGERS:-0}; do
    echo "== $lib stagger=$sg"
    TVC_ROWS_STAGGER=$sg TVC_HIP_LIB=$PWD/tvc_ai_amd/csrc/$lib timeout -k 10 120 python tools/act_bench.py ${ROWS:-16384 65536} 2>&1 | grep -v amdgpu.ids
  done
done
