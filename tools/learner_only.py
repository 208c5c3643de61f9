#!/usr/bin/env python3
"""Runs K SAC updates on a fixed batch (no envs) so a kernel trace shows the learner alone.
usage: rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 tools/learner_only.py [K] [family] [dropout_p]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

from tvc_ai_amd.agent import NativeSAC, sac_cfg

K = int(sys.argv[1]) if len(sys.argv) > 1 else 50
family = int(sys.argv[2]) if len(sys.argv) > 2 else 0
dropout = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
B = 256
d = torch.device("cuda:0")
sac = NativeSAC(sac_cfg(family, batch_size=B, max_act_rows=B, dropout_p=dropout), device=d, seed=1)
g = torch.Generator(device=d).manual_seed(0)
s, s2 = torch.randn((B, 10), device=d, generator=g), torch.randn((B, 10), device=d, generator=g)
a = torch.rand((B, 2), device=d, generator=g) * 2 - 1
r, dn = torch.randn((B,), device=d, generator=g), (torch.rand((B,), device=d, generator=g) < 0.05).float()
e1, e2 = torch.randn((B, 2), device=d, generator=g), torch.randn((B, 2), device=d, generator=g)
for _ in range(10):
    sac.update(s, a, r, s2, dn, e1, e2)
torch.cuda.synchronize()
import time
t = time.perf_counter()
for _ in range(K):
    sac.update(s, a, r, s2, dn, e1, e2)
torch.cuda.synchronize()
print("us/update", (time.perf_counter() - t) / K * 1e6, flush=True)
