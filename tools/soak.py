#!/usr/bin/env python3
"""Soak run of the vector train loop: many steps with domain randomisation + curiosity, then checks that nothing drifted
(finite parameters and losses, bounded replay rewards, stable device memory).  usage: python3 tools/soak.py [steps] [envs] [x3=0|1] [dropout=0|1]
x3 = 1: the acting pass on the split-operand kernel (VecTrainer(acting_x3=True)); dropout = 1: acting in train mode"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

from tvc_ai_amd.env import dr_from_yaml
from tvc_ai_amd.trainer import VecTrainer

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
envs = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
x3 = len(sys.argv) > 3 and sys.argv[3] == "1"
drop = len(sys.argv) > 4 and sys.argv[4] == "1"
tr = VecTrainer(envs, device="cuda:0", family=0, batch_size=256, replay_capacity=1_000_000, seed=7, enable_curiosity=True,
                acting_x3=x3, acting_dropout=drop, defer_join=True, **dr_from_yaml({}, 5))
for _ in range(50):
    tr.step(True)
torch.cuda.synchronize()
mem0 = torch.cuda.memory_allocated()
t0 = time.perf_counter()
worst = 0.0
for k in range(steps):
    tr.step(True)
    if k % 500 == 499:
        torch.cuda.synchronize()
        losses = tr.sac.losses.cpu()
        assert torch.isfinite(losses).all(), (k, losses)
        worst = max(worst, losses[:2].max().item())
        print(f"step {k + 1}: losses {[round(float(x), 3) for x in losses]}  {(k + 1) * envs / (time.perf_counter() - t0) / 1e6:.1f} M env-steps/s",
              flush=True)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
assert torch.isfinite(tr.sac.params).all()
rows, meta = tr.rb.export()
assert torch.isfinite(rows).all() and rows[:, 12].abs().max().item() <= 1000.0 + 1.0  # rewards are clipped to [-1000, 200] (+ bonus)
aux = tr.env.export_state()["aux"]
assert int(aux[:, 0].max()) <= 1000
mem1 = torch.cuda.memory_allocated()
print(f"ok: {steps} steps x {envs} envs in {dt:.1f} s = {steps * envs / dt / 1e6:.1f} M env-steps/s, {steps / dt:.0f} updates/s; "
      f"replay {meta}; device memory {mem0 / 2**20:.0f} -> {mem1 / 2**20:.0f} MiB; largest q-loss seen {worst:.1f}")
tr.close()
