#!/usr/bin/env python3
"""Correctness of a library variant's split-operand kernel against the default f32 kernel of the same library (quick check for
TVC_HIP_LIB experiments: max |mean_x3 - mean_f32| over 65 499 rows must be ~1e-6)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tvc_ai_amd.agent import NativeSAC, sac_cfg
n = 65499
sac = NativeSAC(sac_cfg(0, batch_size=64, max_act_rows=65536), device="cuda:0", seed=13)
ob, ep = torch.randn(n, 10, device="cuda") * 0.5, torch.randn(n, 2, device="cuda")
a = [t.clone() for t in sac.act(ob, ep)]
b = [t.clone() for t in sac.act(ob, ep, x3=True)]
d = max(float((x - y).abs().max()) for x, y in zip(a[1:], b[1:]))
print("max |x3 - f32| over mean / log_std:", d, "OK" if d < 5e-5 else "MISMATCH")
sys.exit(0 if d < 5e-5 else 1)
