#!/bin/bash
# timing build of the library with in-kernel stamps in the split-K GEMM (tools/gemm_stamps.py); the product build has none
cd "$(dirname "$0")/.." && /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -fno-gpu-rdc -ffp-contract=fast -fno-slp-vectorize \
  -DTVC_GEMM_STAMPS -I include -I tvc_ai_amd/csrc -o tvc_ai_amd/csrc/libtvc_hip_stamps.so tvc_ai_amd/csrc/tvc_env.hip tvc_ai_amd/csrc/tvc_sac.hip tvc_ai_amd/csrc/tvc_replay.hip 2>&1 | grep -E "error" -A4
ls -la tvc_ai_amd/csrc/libtvc_hip_stamps.so
