"""Integrator with the reference-exact 1000-entry reward history, rings FULL (1010 steps in): env_step_kernel<W1000, DR> at N envs.
The scan of the env's ring (4 KB per env-step) is a chain of 1000 / TVC_RING_BATCH round trips.  One JSON line per N.
usage: env_ring_bench.py [N ...]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tvc_ai_amd import VecRocketTVCEnv
from tvc_ai_amd.env import dr_from_yaml

dev = torch.device("cuda:0")
for n in [int(a) for a in sys.argv[1:]] or [4096, 65536, 1 << 20]:
    env = VecRocketTVCEnv(n, device=dev, seed=7, distinct_window=1000, **dr_from_yaml({}, 5))
    env.enable_episode_stats()
    env.reset()
    acts = (torch.rand((8, n, 2), device=dev) * 2 - 1).contiguous()
    for k in range(1010):
        env.step(acts[k % 8])
    us = bench.graph_time_us(lambda k: env.step(acts[k % 8]), 50, dev)
    per_env = bench.ENV_STEP_BYTES_DR + 8 + 4 * 1000
    gbs = per_env * n / (us * 1e-6) / 1e9
    print(json.dumps({"envs": n, "window": 1000, "launch_us": us, "algorithmic_bytes_per_env_step": per_env, "GB/s": gbs,
                      "frac_of_8TBs": gbs / 8000.0}), flush=True)
    env.close()
