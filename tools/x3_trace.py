#!/usr/bin/env python3
"""Per-pass cycle stamps of wave 0 in the split-operand acting kernel (library built with -DAR_TRACE, TVC_HIP_LIB pointing at it).
usage (GPU box): TVC_HIP_LIB=tvc_ai_amd/csrc/libtvc_hip_trace.so python tools/x3_trace.py [rows=16384] [x3=1]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tvc_ai_amd.agent import NativeSAC, sac_cfg  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
x3 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
sac = NativeSAC(sac_cfg(0, batch_size=256, max_act_rows=n), device="cuda:0", seed=2)
ob = torch.randn(n, 10, device="cuda:0")
nwg = (n + 63) // 64
STRIDE = 96
buf = (C.c_uint64 * (nwg * STRIDE))()
rc = sac.L.tvc_debug_rows_stamps(sac._h, ob.data_ptr(), n, 6, 16 if x3 else 0, buf, torch.cuda.current_stream().cuda_stream)
assert rc == 0, sac.L.tvc_last_error()
v = np.frombuffer(buf, dtype=np.uint64).reshape(nwg, STRIDE).astype(np.int64)
t0 = v[:, 0]
st = v[:, 6:]
nst = int((st[0] > 0).sum())
d = np.diff(np.concatenate([t0[:, None], st[:, :nst]], axis=1), axis=1)
med = np.median(d, axis=0)
print(f"rows {n} x3={x3}: {nst} stamps, total {np.median(st[:, nst - 1] - t0):.0f} cycles (median over {nwg} workgroups), lifetime {np.median(v[:, 2] - v[:, 0]):.0f}")
for i, m in enumerate(med):
    print(f"  seg {i:2d}: {m:9.0f} cycles")
