#!/usr/bin/env python3
"""Where does a batch-256 GEMM launch spend its ~6 us?  Timing build of the library (-DTVC_GEMM_STAMPS: s_memtime stamps of
thread 0 of every workgroup at: entry, operands landed, MFMAs done (LDS writes issued), partial-sum barrier passed, stores issued,
stores acknowledged) on a chain of dependent launches; prints per-phase medians in shader cycles and the in-launch span.
usage: gemm_stamps.py [M N K]      (needs tvc_ai_amd/csrc/libtvc_hip_stamps.so: tools/build_stamps.sh)"""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["TVC_HIP_LIB"] = os.path.join(ROOT, "tvc_ai_amd", "csrc", "libtvc_hip_stamps.so")
sys.path.insert(0, ROOT)
import numpy as np
import torch
from tvc_ai_amd import _native as nat

M, N, K = [int(x) for x in sys.argv[1:4]] if len(sys.argv) > 3 else (512, 256, 256)
lib = nat.load()
dev = torch.device("cuda:0")
nblk = (N // 32) * (M // 32)
st = torch.zeros(nblk * 8, dtype=torch.int64, device=dev)
lib.tvc_debug_set_gemm_stamps.argtypes = [C.c_void_p]
lib.tvc_debug_set_gemm_stamps.restype = None
bufs = [torch.randn(M, max(N, K), device=dev) * 0.01 for _ in range(2)]
W = torch.randn(N, K, device=dev) / K ** 0.5
b = torch.zeros(N, device=dev)
s = torch.cuda.Stream(dev)
with torch.cuda.stream(s):
    def fn(k):
        nat.check(lib.tvc_nn_linear_forward(bufs[k & 1].data_ptr(), W.data_ptr(), b.data_ptr(), bufs[1 - (k & 1)].data_ptr(), M, N, K, 0, 3,
                                            torch.cuda.current_stream().cuda_stream))
    lib.tvc_debug_set_gemm_stamps(None)
    for k in range(4):
        fn(k)
    g = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    lib.tvc_debug_set_gemm_stamps(st.data_ptr())
    with torch.cuda.graph(g, stream=s):
        for k in range(32):
            fn(k)
    for _ in range(5):
        g.replay()
torch.cuda.synchronize()
v = st.cpu().numpy().reshape(nblk, 8).astype(np.float64)
t0 = v[:, 0].min()
names = ["entry", "prefetch_landed", "operands_landed", "mfma_done", "barrier_passed", "lds_summed", "stores_issued", "stores_acked"]
rep = {"shape": [M, N, K], "workgroups": nblk,
       "median_cycles_since_first_entry": {n: float(np.median(v[:, i] - t0)) for i, n in enumerate(names)},
       "max_cycles_since_first_entry": {n: float((v[:, i] - t0).max()) for i, n in enumerate(names)},
       "median_phase_cycles": {names[i + 1]: float(np.median(v[:, i + 1] - v[:, i])) for i in range(7)},
       "entry_spread_cycles": float(v[:, 0].max() - t0)}
print(json.dumps(rep, indent=1))
