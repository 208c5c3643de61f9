"""Diagnostic: per-block step times of the segment-graph loop for the FIRST trainer of a process (an intermittent 3.7 ms/step was seen)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from types import SimpleNamespace
from tvc_ai_amd import trainer
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
mode = sys.argv[2] if len(sys.argv) > 2 else "segments"
args = SimpleNamespace(family=0, envs_per_gpu=n, dr_stage=5, exact_reward=False, shipped_acting=False, updates_per_step=1, no_overlap=False,
                       share_cus="auto", share_rows=-1, reward_window=0, acting_dropout=False, prefill_steps=int(os.environ.get("PREFILL", "1000")))
dev = torch.device("cuda:0")
res = trainer.bench_train(args, 1, 0, dev, n_envs=n)
t = res["trainer"]; fn = res["step_fn"]
if mode == "segments":
    seg = t.capture_segments(); fn = lambda k: seg()
for blk in range(12):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(25):
        fn(k)
    torch.cuda.synchronize()
    print(mode, n, "block", blk, round((time.perf_counter() - t0) / 25 * 1e3, 3), "ms/step", flush=True)
t.close()
