"""Kernel A/B timing: env_step at a small (graph) and a large (eager) size for the lib in $TVC_HIP_LIB."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tvc_ai_amd import VecRocketTVCEnv

def time_graph(n, K=300, **kw):
    env = VecRocketTVCEnv(n, **kw); env.reset()
    acts = (torch.rand((16, n, 2), device="cuda") * 2 - 1).contiguous()
    for k in range(20): env.step(acts[k % 16])
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for k in range(K): env.step(acts[k % 16])
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / K * 1e3)
    env.close()
    return best

def time_eager(n, K=30, **kw):
    env = VecRocketTVCEnv(n, **kw); env.reset()
    acts = (torch.rand((4, n, 2), device="cuda") * 2 - 1).contiguous()
    for k in range(60): env.step(acts[k % 4])   # spread the envs over episode phases
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(K): env.step(acts[k % 4])
    e1.record(); torch.cuda.synchronize()
    env.close()
    return e0.elapsed_time(e1) / K * 1e3

tag = os.path.basename(os.environ.get("TVC_HIP_LIB", "default"))
small = time_graph(8192)
big = time_eager(1 << 22)
small_dr = time_graph(8192, dr_enabled=1, dr_mass_var=0.3, dr_thrust_std=0.2, dr_cg_max=0.1, dr_wind_std=3.0, dr_obs_noise_std=0.02)
print(f"{tag:28s} N=8192: {small:7.2f} us/step   N=4M: {big:8.1f} us/step = {(1<<22)/big/1e3:6.2f} G env-steps/s, {242*(1<<22)/big/1e3:7.1f} GB/s algorithmic   N=8192+DR: {small_dr:7.2f} us", flush=True)
