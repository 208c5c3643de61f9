"""Diagnostic: error growth GPU(fp32) vs oracle(fp64) around ground contact."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import envoracle as eo
from tvc_ai_amd import VecRocketTVCEnv

n = 256
rng = np.random.default_rng(11)
env = VecRocketTVCEnv(n, contact=1, auto_reset=0)
env.reset()
vec = eo.OracleVec(n, contact=1, auto_reset=0, distinct_window=10)
acts = rng.uniform(-0.3, 0.3, (80, n, 2)).astype(np.float32)
for t in range(80):
    env.step(torch.from_numpy(acts[t]).cuda())
    vec.step(acts[t].astype(np.float64))
    g = env.export_state()["dyn"].cpu().numpy().astype(np.float64)
    o = vec.state13()
    err = np.abs(g - o) / np.maximum(1, np.abs(o))
    i = np.unravel_index(np.argmax(err), err.shape)
    if t >= 28:
        print(t, "max err %.2e env %d comp %d  z=%.4f  median-env-max %.2e  n>1e-4: %d" % (err.max(), i[0], i[1], o[i[0], 2], np.median(err.max(axis=1)), (err.max(axis=1) > 1e-4).sum()))
