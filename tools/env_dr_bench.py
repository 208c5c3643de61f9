"""Integrator in the bandwidth regime: env_step_kernel at N envs (default 4 M), nominal and domain-randomised (stage 5), episode
statistics off / on, 10-entry window.  Prints one JSON line per variant: launch us, GB/s on the algorithmic bytes, fraction of 8 TB/s.
usage: env_dr_bench.py [N]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tvc_ai_amd import VecRocketTVCEnv
from tvc_ai_amd.env import dr_from_yaml

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 22
dev = torch.device("cuda:0")
for name, over, stats, per_env in (("nominal", {}, False, bench.ENV_STEP_BYTES),
                                   ("dr5", dr_from_yaml({}, 5), False, bench.ENV_STEP_BYTES_DR),
                                   ("dr5+stats", dr_from_yaml({}, 5), True, bench.ENV_STEP_BYTES_DR + 8),
                                   ("dr5 no-noise", {**dr_from_yaml({}, 5), "dr_obs_noise_std": 0.0}, False, bench.ENV_STEP_BYTES_DR)):
    env = VecRocketTVCEnv(n, device=dev, seed=7, **over)
    if stats:
        env.enable_episode_stats()
    env.reset()
    acts = (torch.rand((8, n, 2), device=dev) * 2 - 1).contiguous()
    for k in range(40):
        env.step(acts[k % 8])
    us = bench.graph_time_us(lambda k: env.step(acts[k % 8]), 50, dev)
    gbs = per_env * n / (us * 1e-6) / 1e9
    print(json.dumps({"variant": name, "envs": n, "launch_us": us, "algorithmic_bytes_per_env_step": per_env, "GB/s": gbs,
                      "frac_of_8TBs": gbs / 8000.0}), flush=True)
    env.close()
