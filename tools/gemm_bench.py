"""Bare fused-Linear kernel benchmark + numerics vs torch (fp32).  usage: gemm_bench.py [lib.so ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tvc_ai_amd import _native as nat
L = nat.load()
st = lambda: torch.cuda.current_stream().cuda_stream
shapes = [(65536, 256, 256), (65536, 512, 256), (65536, 256, 512), (65536, 512, 512), (8192, 256, 256), (8192, 512, 256), (256, 256, 256), (256, 512, 256), (256, 256, 512)]
for (M, N, K) in shapes:
    X = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") / K ** 0.5; b = torch.randn(N, device="cuda")
    Y = torch.empty(M, N, device="cuda")
    ref = torch.nn.functional.linear(X, W, b)
    line = f"M={M:6d} N={N:4d} K={K:4d}: "
    for v in (1, 3):
        if v == 3 and M > 1024: continue
        nat.check(L.tvc_nn_linear_forward(X.data_ptr(), W.data_ptr(), b.data_ptr(), Y.data_ptr(), M, N, K, 0, v, st()))
        torch.cuda.synchronize()
        err = (Y - ref).abs().max().item()
        reps = 20
        g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            with torch.cuda.graph(g, stream=s):
                for _ in range(reps):
                    L.tvc_nn_linear_forward(X.data_ptr(), W.data_ptr(), b.data_ptr(), Y.data_ptr(), M, N, K, 0, v, st())
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / reps * 1e3)
        line += f" v{v}: {best:7.1f} us {2*M*N*K/best/1e6:6.1f} TF err {err:.1e} |"
    if M > 1024 and N in (256, 512):  # fused Linear (+GELU for N = 512) + residual + LayerNorm (tvc_nn_linear_ln_forward)
        R = torch.randn(M, N, device="cuda"); gam = torch.ones(N, device="cuda"); bet = torch.zeros(N, device="cuda")
        act = 0 if N == 256 else 1
        g = torch.cuda.CUDAGraph(); s_ = torch.cuda.Stream(); s_.wait_stream(torch.cuda.current_stream())
        call = lambda: L.tvc_nn_linear_ln_forward(X.data_ptr(), W.data_ptr(), b.data_ptr(), R.data_ptr(), gam.data_ptr(), bet.data_ptr(),
                                                  Y.data_ptr(), M, N, K, act, st())
        with torch.cuda.stream(s_):
            call()
            with torch.cuda.graph(g, stream=s_):
                for _ in range(20):
                    call()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
        line += f" +res+LN: {best:7.1f} us {2*M*N*K/best/1e6:6.1f} TF |"
    if M > 1024:  # the vendor library on the same shape (torch.nn.functional.linear -> hipBLASLt / rocBLAS), bias included
        g = torch.cuda.CUDAGraph(); s_ = torch.cuda.Stream(); s_.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s_):
            for _ in range(3):
                torch.nn.functional.linear(X, W, b, ) if False else torch.addmm(b, X, W.t(), out=Y)
            with torch.cuda.graph(g, stream=s_):
                for _ in range(20):
                    torch.addmm(b, X, W.t(), out=Y)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
        line += f" torch.addmm: {best:7.1f} us {2*M*N*K/best/1e6:6.1f} TF |"
    print(line, flush=True)
