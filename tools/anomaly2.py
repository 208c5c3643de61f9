"""Diagnostic for the 3.6 ms/step seen by the SECOND 4 096-env trainer of a process (bench.py shard_sizes: segment graphs, then eager).
usage: anomaly2.py <variant>   variants: base | nocapture | gc | prio0 | noreplay | other_n | eager_first"""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from types import SimpleNamespace
variant = sys.argv[1] if len(sys.argv) > 1 else "base"
from tvc_ai_amd import trainer

dev = torch.device("cuda:0")
def mk(n):
    args = SimpleNamespace(family=0, envs_per_gpu=n, dr_stage=5, exact_reward=False, shipped_acting=False, updates_per_step=1, no_overlap=False,
                           share_cus="auto", share_rows=-1, reward_window=0, acting_dropout=False, prefill_steps=1000)
    return trainer.bench_train(args, 1, 0, dev, n_envs=n)
def timeit(fn, steps=150):
    for k in range(30): fn(k)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(steps): fn(k)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / steps * 1e3

if variant == "eager_first":
    r = mk(4096); print(variant, "eager #1", round(timeit(r["step_fn"]), 3), flush=True); r["trainer"].close()
    r = mk(4096); print(variant, "eager #2", round(timeit(r["step_fn"]), 3), flush=True); r["trainer"].close()
    sys.exit(0)
r = mk(4096 if variant != "other_n" else 2048)
t = r["trainer"]
if variant == "nocapture":
    print(variant, "first (eager)", round(timeit(r["step_fn"]), 3), flush=True)
else:
    seg = t.capture_segments()
    if variant != "noreplay":
        print(variant, "first (segments)", round(timeit(lambda k: seg()), 3), flush=True)
t.close()
if variant == "gc":
    del seg, t, r; gc.collect(); torch.cuda.empty_cache()
if variant == "prio0":
    os.environ["TVC_SIDE_PRIORITY"] = "0"
r2 = mk(4096)
print(variant, "second (eager)", round(timeit(r2["step_fn"]), 3), flush=True)
print(variant, "second again", round(timeit(r2["step_fn"]), 3), flush=True)
r2["trainer"].close()
