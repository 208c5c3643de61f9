"""Workload for rocprofv3 --pmc passes: a few eager env_step launches at two sizes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tvc_ai_amd import VecRocketTVCEnv
sizes = [int(x) for x in (sys.argv[1:] or ["8192", "4194304"])]
for n in sizes:
    env = VecRocketTVCEnv(n); env.reset()
    acts = (torch.rand((4, n, 2), device="cuda") * 2 - 1).contiguous()
    for k in range(40):
        env.step(acts[k % 4])
    torch.cuda.synchronize()
    env.close()
