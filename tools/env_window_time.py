"""env_step launch time at 65 536 envs (DR stage 5, episode statistics) with the 10-entry and the 1000-entry reward history"""
import importlib.util, os, torch
spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
for w in (10, 1000):
    r = b.integrator_roofline(65536, torch.device("cuda:0"), dr_stage=5, stats=True, window=w)
    print(w, round(r["launch_us"], 2))
