#!/usr/bin/env python3
"""SAC update alone (batch 256, reference shapes, train-mode dropout 0.1): hipGraph of 20 updates, HIP events on its stream, best
of 5 -- the `sac_learner_only` leg of bench.py on its own.  TVC_FOLD_LN=0 runs the LayerNorms as their own launches (A/B).
usage: update_bench.py [family] [dropout_p]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tvc_ai_amd.agent import NativeSAC, sac_cfg

family = int(sys.argv[1]) if len(sys.argv) > 1 else 0
p = float(sys.argv[2]) if len(sys.argv) > 2 else (0.1 if family == 0 else 0.0)
dev = torch.device("cuda:0")
B = 256
sac = NativeSAC(sac_cfg(family, batch_size=B, max_act_rows=B, dropout_p=p), device=dev, seed=1)
bt = (torch.randn(B, 10, device=dev), torch.rand(B, 2, device=dev) * 2 - 1, torch.randn(B, device=dev),
      torch.randn(B, 10, device=dev), torch.zeros(B, device=dev), torch.randn(B, 2, device=dev), torch.randn(B, 2, device=dev))
us = bench.graph_time_us(lambda k: sac.update(*bt), 20, dev)
print(json.dumps({"family": family, "dropout_p": p, "us_per_update": us, "updates_per_s": 1e6 / us,
                  "fold_ln": os.environ.get("TVC_FOLD_LN", "0"), "ln_tail": os.environ.get("TVC_LN_TAIL", "1"), "adam_steps": sac.adam_steps(),
                  "losses": sac.losses.cpu().tolist()}), flush=True)
sac.close()
