#!/bin/bash
# acting bench of the f32-MFMA one-launch kernel over library variants (TVC_HIP_LIB): f32_variants.sh <out> <name> ...   ("base" = the default library)
out=$1; shift
mkdir -p $(dirname $out); : > $out
for v in "$@"; do
  lib=tvc_ai_amd/csrc/libtvc_hip_$v.so; [ $v = base ] && lib=tvc_ai_amd/csrc/libtvc_hip.so
  echo "== $v" >> $out
  TVC_HIP_LIB=$lib timeout -k 10 120 python tools/act_bench.py 16384 65536 2>&1 | grep "^rows" >> $out || exit 1
  TVC_HIP_LIB=$lib TVC_ACT_SHARE=1 timeout -k 10 120 python tools/act_bench.py 32768 2>&1 | grep "^rows" >> $out || exit 1
done
cat $out
