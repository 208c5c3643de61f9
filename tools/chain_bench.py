#!/usr/bin/env python3
"""What does ONE dependent launch of the update's GEMM cost?  hipGraph chains of L launches, HIP events, us per launch:
  trivial      : x.add_(1) on one element (the launch floor)
  skinny MxNxK : tvc_nn_linear_forward (split-K kernel) ping-ponging two activation buffers, so each launch reads what the
                 previous one wrote
usage: chain_bench.py [L=64]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tvc_ai_amd import _native as nat

L_ = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda:0")
lib = nat.load()


def chain(fn, L=L_, R=50):
    s = torch.cuda.Stream(dev)
    with torch.cuda.stream(s):
        for k in range(4):
            fn(k)
        g = torch.cuda.CUDAGraph()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for k in range(L):
                fn(k)
        for _ in range(3):
            g.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(R):
            g.replay()
        e1.record(s)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / R / L


out = {"HIP_FORCE_DEV_KERNARG": os.environ.get("HIP_FORCE_DEV_KERNARG")}
x = torch.zeros(1, device=dev)
out["trivial"] = chain(lambda k: x.add_(1.0))
for (M, N, K) in ((512, 256, 256), (256, 256, 256), (512, 512, 512), (512, 256, 512)):
    bufs = [torch.randn(M, max(N, K), device=dev) * 0.01 for _ in range(2)]
    W = torch.randn(N, K, device=dev) / K ** 0.5
    b = torch.zeros(N, device=dev)

    def fn(k, M=M, N=N, K=K, bufs=bufs, W=W, b=b):
        nat.check(lib.tvc_nn_linear_forward(bufs[k & 1].data_ptr(), W.data_ptr(), b.data_ptr(), bufs[1 - (k & 1)].data_ptr(), M, N, K, 0, 3,
                                            torch.cuda.current_stream().cuda_stream))
    if N == K:
        out[f"skinny_{M}x{N}x{K}"] = chain(fn)
    else:  # not chainable in place (shapes differ): same input every time
        X = torch.randn(M, K, device=dev)
        Y = torch.empty(M, N, device=dev)
        out[f"skinny_{M}x{N}x{K}_indep_input"] = chain(lambda k: nat.check(lib.tvc_nn_linear_forward(
            X.data_ptr(), W.data_ptr(), b.data_ptr(), Y.data_ptr(), M, N, K, 0, 3, torch.cuda.current_stream().cuda_stream)))
big = torch.zeros(512, 256, device=dev)
out["elementwise_512x256"] = chain(lambda k: big.add_(1.0))
print(json.dumps(out), flush=True)
