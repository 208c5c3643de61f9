#!/usr/bin/env python3
"""Golden vectors for the two small acting-path MLPs of the reference, from the reference's own code:
  CuriosityModule.compute_intrinsic_reward  (env/enhanced_rocket_tvc_env.py:226-269)  -- untrained forward model
  SafetyLayer.forward                       (agent/multi_algorithm_agent.py:287-351)
Weights come from the numpy recipe of gen_sac_golden.fill_params (nothing but outputs is stored).
Build container only (needs /root/reference).  Output: tests/golden/aux_ref.npz"""
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
SEED = 777


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


def make_inputs(rng, n=256):
    s = rng.standard_normal((n, 10)).astype(np.float32) * 0.5
    q = rng.standard_normal((n, 4)) * np.array([0.3, 0.3, 0.3, 1.0])
    s[:, :4] = (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(np.float32)
    s[:, 4:7] *= 6.0          # some rows beyond max_angular_velocity = 5
    a = (rng.standard_normal((n, 2)) * 0.7).astype(np.float32)  # some rows with |a| > 1
    s2 = (s + 0.05 * rng.standard_normal((n, 10))).astype(np.float32)
    return s, a, s2


def main():
    sys.dont_write_bytecode = True
    rec = _load("gen_sac_golden", os.path.join(HERE, "gen_sac_golden.py"))
    sys.path.insert(0, HERE)
    genv = _load("gen_env_golden", os.path.join(HERE, "gen_env_golden.py"))
    envmod = genv.load_reference_env_module()
    sys.path.insert(0, REF)
    from agent.multi_algorithm_agent import SafetyLayer, SafetyConstraints

    rng = np.random.default_rng(SEED)
    out = {}
    # --- curiosity: forward model Linear(10,256)-ReLU-Linear(256,256)-ReLU-Linear(256,8)
    cm = envmod.CuriosityModule(obs_dim=8, action_dim=2)
    named = [(n, tuple(p.shape)) for n, p in cm.forward_model.named_parameters()]
    vals = rec.fill_params(named, rng)
    with torch.no_grad():
        for n, p in cm.forward_model.named_parameters():
            p.copy_(torch.from_numpy(vals[n]))
    s, a, s2 = make_inputs(rng)
    out["cur_reward"] = np.array([cm.compute_intrinsic_reward(s[i, :8], np.clip(a[i], -1, 1), s2[i, :8]) for i in range(64)])
    # --- safety layer
    sl = SafetyLayer(2, SafetyConstraints())
    named = [(n, tuple(p.shape)) for n, p in sl.safety_net.named_parameters()]
    vals = rec.fill_params(named, rng)
    with torch.no_grad():
        for n, p in sl.safety_net.named_parameters():
            p.copy_(torch.from_numpy(vals[n]))
    s, a, s2 = make_inputs(rng)
    with torch.no_grad():
        corrected = sl(torch.from_numpy(s), torch.from_numpy(a))
        out["safety_out"] = torch.clamp(corrected, -1.0, 1.0).numpy()  # get_action clamps afterwards (agent/...:789)
        none_viol = sl(torch.from_numpy(s[:4] * 0 + np.array([0, 0, 0, 1, 0, 0, 0, 1, 0, 0], np.float32)),
                       torch.from_numpy(a[:4] * 0.1))
        out["safety_noviol"] = none_viol.numpy()
    np.savez_compressed(os.path.join(HERE, "aux_ref.npz"), **out)
    print({k: v.shape for k, v in out.items()}, "violating rows:", int((np.abs(out["safety_out"] - np.clip(a, -1, 1)).max(1) > 0).sum()))


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("needs /root/reference (build container only)")
    main()
