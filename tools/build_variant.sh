#!/bin/bash
# build a variant of the library: build_variant.sh <name> <extra hipcc flags...>  -> tvc_ai_amd/csrc/libtvc_hip_<name>.so (use with TVC_HIP_LIB)
cd "$(dirname "$0")/.." && name=$1 && shift && /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -fno-gpu-rdc -ffp-contract=fast -fno-slp-vectorize \
  "$@" -I include -I tvc_ai_amd/csrc -o tvc_ai_amd/csrc/libtvc_hip_$name.so tvc_ai_amd/csrc/tvc_env.hip tvc_ai_amd/csrc/tvc_sac.hip tvc_ai_amd/csrc/tvc_replay.hip 2>&1 | grep -E "error" -A4
ls -la tvc_ai_amd/csrc/libtvc_hip_$name.so
