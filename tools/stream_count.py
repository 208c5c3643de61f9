"""Does the train loop slow down once a process has touched many HIP streams?  (ROCm multiplexes streams onto a few hardware queues;
the loop needs the learner's stream and the main stream on DIFFERENT queues.)
usage: stream_count.py <n_normal_streams> <n_high_priority_streams> [n_envs]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from types import SimpleNamespace
from tvc_ai_amd import trainer
dev = torch.device("cuda:0")
nn, nh = int(sys.argv[1]), int(sys.argv[2])
n = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
x = torch.zeros(1024, device=dev)
keep = []
for k in range(nn):
    s = torch.cuda.Stream(dev); keep.append(s)
    with torch.cuda.stream(s): x.add_(1.0)
for k in range(nh):
    s = torch.cuda.Stream(dev, priority=-1); keep.append(s)
    with torch.cuda.stream(s): x.add_(1.0)
torch.cuda.synchronize()
args = SimpleNamespace(family=0, envs_per_gpu=n, dr_stage=5, exact_reward=False, shipped_acting=False, updates_per_step=1, no_overlap=False,
                       share_cus="auto", share_rows=-1, reward_window=0, acting_dropout=False, prefill_steps=1000)
r = trainer.bench_train(args, 1, 0, dev, n_envs=n)
fn = r["step_fn"]
for k in range(30): fn(k)
torch.cuda.synchronize(); t0 = time.perf_counter()
for k in range(200): fn(k)
torch.cuda.synchronize()
print(f"streams touched before: {nn} normal + {nh} high priority; GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES', 'default')}: "
      f"{(time.perf_counter() - t0) / 200 * 1e3:.3f} ms/step at {n} envs; tuning {r['extra']['sac']['learner_stream']}", flush=True)
r["trainer"].close()
