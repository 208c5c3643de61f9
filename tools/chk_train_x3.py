#!/usr/bin/env python3
"""Diagnostic: the train loop acting in train mode with the f32 kernel (actor_split_kernel<true>) and with the split-operand kernel
(actor_x3_kernel<true>): first-step actions at equal call counters, then loss / reward statistics over a few hundred steps."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tvc_ai_amd.trainer import VecTrainer
from tvc_ai_amd.env import dr_from_yaml
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
acts = []
for x3 in (False, True):
    tr = VecTrainer(n, family=0, batch_size=256, replay_capacity=1_000_000, seed=7, acting_dropout=True, acting_x3=x3, share_rows=0,
                    defer_join=True, enable_curiosity=True, **dr_from_yaml({}, 5))
    tr.sac.set_act_counter(0)
    torch.manual_seed(3)
    tr.step(True)
    torch.cuda.synchronize()
    acts.append(tr.act.clone())
    rs = []
    for k in range(600):
        tr.step(True)
        if k % 200 == 199:
            torch.cuda.synchronize()
            rows, meta = tr.rb.export()
            print(f"x3={x3} step {k+1}: losses {[round(float(v),2) for v in tr.sac.losses.cpu()]} replay reward mean {float(rows[:,12].mean()):.2f} min {float(rows[:,12].min()):.1f} "
                  f"done frac {float(rows[:,23].mean()) if rows.shape[1] > 23 else -1:.4f} |act| mean {float(tr.act.abs().mean()):.3f}", flush=True)
    tr.close()
print("first-step max |act_x3 - act_f32|", float((acts[0] - acts[1]).abs().max()), "mean |act|", float(acts[0].abs().mean()))
