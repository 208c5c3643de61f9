#!/usr/bin/env python3
"""Where wave 0 of each workgroup of the split-operand acting kernel waits at the ring's slot boundaries (library built with
-DAR_TRACE -DX3_WAITS): cycles in the copy wait (vmcnt), the LDS wait (lgkmcnt), the barrier and the copy issue of x3_next, against
the workgroup's lifetime; plus the per-pass stamps.
usage (GPU box): TVC_HIP_LIB=tvc_ai_amd/csrc/libtvc_hip_waits.so python tools/x3_waits.py [rows=65536]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tvc_ai_amd.agent import NativeSAC, sac_cfg  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
sac = NativeSAC(sac_cfg(0, batch_size=256, max_act_rows=n), device="cuda:0", seed=2)
ob = torch.randn(n, 10, device="cuda:0")
nwg = (n + 63) // 64
buf = (C.c_uint64 * (nwg * 96))()
rc = sac.L.tvc_debug_rows_stamps(sac._h, ob.data_ptr(), n, 6, 16, buf, torch.cuda.current_stream().cuda_stream)
assert rc == 0, sac.L.tvc_last_error()
v = np.frombuffer(buf, dtype=np.uint64).reshape(nwg, 96).astype(np.int64)
life = v[:, 2] - v[:, 0]
w = v[:, 90:94]
print(f"rows {n}: {nwg} workgroups, lifetime median {np.median(life):.0f} cycles")
for name, col in zip(("copy wait (vmcnt)", "LDS wait (lgkmcnt)", "barrier", "copy issue"), range(4)):
    print(f"  {name:20s} median {np.median(w[:, col]):10.0f} cycles = {100 * np.median(w[:, col] / life):5.1f} % of the lifetime")
st = v[:, 6:90]
nst = int((st[0] > 0).sum())
d = np.diff(np.concatenate([v[:, 0:1], st[:, :nst]], axis=1), axis=1)
med = np.median(d, axis=0)
print("  per-pass stamps (median cycles):", " ".join(f"{m:.0f}" for m in med))
