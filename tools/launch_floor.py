#!/usr/bin/env python3
"""Per-launch floor on this GPU: a trivial dependent kernel chain, eager and replayed from a hipGraph.
usage: python3 tools/launch_floor.py"""
import time

import torch

d = torch.device("cuda:0")
x = torch.zeros(64, device=d)
N = 200


def chain():
    for _ in range(N):
        x.add_(1.0)


chain()
torch.cuda.synchronize()
t = time.perf_counter()
chain()
torch.cuda.synchronize()
print("eager  us/launch", (time.perf_counter() - t) / N * 1e6)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    chain()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        chain()
g.replay()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    g.replay()
e1.record()
torch.cuda.synchronize()
print("graph  us/launch", e0.elapsed_time(e1) * 1e3 / (5 * N))
