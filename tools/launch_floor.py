#!/usr/bin/env python3
"""Floor of a dependent launch chain on this GPU: a hipGraph of L trivial, serially dependent kernels (one-element add), timed
with HIP events over R replays.  One SAC update at batch 256 is 97 dependent launches (profiles/r01_h_learner_update_timeline.md):
this measures what those 97 launches cost when the kernels do nothing.
usage (GPU box): python tools/launch_floor.py [L=97] [R=200]"""
import json
import sys

import torch

L = int(sys.argv[1]) if len(sys.argv) > 1 else 97
R = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dev = torch.device("cuda:0")
x = torch.zeros(1, device=dev)
big = torch.zeros(512, 256, device=dev)
out = {}
for name, fn in (("one_element_add", lambda: x.add_(1.0)), ("512x256_elementwise", lambda: big.add_(1.0))):
    s = torch.cuda.Stream(dev)
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
        g = torch.cuda.CUDAGraph()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(L):
                fn()
        for _ in range(5):
            g.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(R):
            g.replay()
        e1.record(s)
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / R
    out[name] = {"launches": L, "us_per_chain": us, "us_per_launch": us / L}
print(json.dumps(out))
