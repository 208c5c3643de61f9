#!/usr/bin/env python3
"""Golden vectors for the SAC learner half, taken from the reference's own code.

Runs ONLY in the build container (needs /root/reference).  Imports
agent/multi_algorithm_agent.py unmodified (torch + numpy only) and drives
  MultiAlgorithmAgent._create_sac_agent  (agent/multi_algorithm_agent.py:587-627)
  TransformerPolicyNetwork.forward       (:192-227)
  MultiAlgorithmAgent.update / _update_sac (:868-912, :950-1016)
  PhysicsInformedLoss.forward            (:236-285)
with the nets in .eval() (dropout off) and the two Gaussian draws of each update replaced by
pre-generated eps tensors (torch.distributions.Normal.sample/rsample patched in this process only).

No weights are stored: both sides fill every parameter from numpy's PCG64 stream (see fill_params),
so the fixture holds only the recipe (names, shapes, seed) and the reference's OUTPUTS.
Output: tests/golden/sac_ref.npz (+ sac_ref_meta.json).
"""
import json
import os
import sys

import numpy as np
import torch
import yaml

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
SEED = 20251004
B = 256


def fill_params(named_params, rng):
    """Deterministic parameter values from a numpy Generator, in iteration order.
    matrices: N(0,1)/sqrt(fan_in); 1-D '*weight' (LayerNorm gamma): 1 + 0.1 N(0,1); other 1-D: 0.05 N(0,1)."""
    out = {}
    for name, shape in named_params:
        if len(shape) >= 2:
            v = rng.standard_normal(shape) / np.sqrt(shape[1])
        elif name.endswith("weight"):
            v = 1.0 + 0.1 * rng.standard_normal(shape)
        else:
            v = 0.05 * rng.standard_normal(shape)
        out[name] = v.astype(np.float32)
    return out


def make_batch(rng, b=B):
    s = rng.standard_normal((b, 10)) * 0.5
    s[:, :4] /= np.linalg.norm(s[:, :4], axis=1, keepdims=True)
    s[:, 7:10] = rng.uniform(0, 1, (b, 3))
    s2 = s + 0.05 * rng.standard_normal((b, 10))
    a = rng.uniform(-1, 1, (b, 2))
    r = rng.normal(50.0, 30.0, b)
    d = (rng.uniform(0, 1, b) < 0.1).astype(np.float64)
    return [x.astype(np.float32) for x in (s, a, r, s2, d)]


def tensor_digest(t):
    a = t.detach().cpu().numpy().astype(np.float64).ravel()
    return np.array([a.sum(), np.abs(a).sum(), (a * a).sum(), a[0], a[len(a) // 2], a[-1]])


def main():
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    from agent.multi_algorithm_agent import MultiAlgorithmAgent  # the reference, unmodified
    import torch.distributions as D

    cfg = yaml.safe_load(open(os.path.join(REF, "config", "config.yaml")))
    cfg["algorithms"]["ppo"]["enabled"] = False
    cfg["algorithms"]["td3"]["enabled"] = False
    cfg["hierarchical_rl"]["enabled"] = False
    cfg["safety"]["safety_layer"]["enabled"] = False
    cfg["hardware"] = {"device": "cpu"}
    torch.manual_seed(0)
    agent = MultiAlgorithmAgent(10, 2, cfg)
    agent.device = torch.device("cpu")
    sac = agent.algorithms["sac"]
    nets = {k: sac[k] for k in ("policy", "q1", "q2", "target_q1", "target_q2")}

    rng = np.random.default_rng(SEED)
    meta = {"seed": SEED, "batch": B, "nets": {}}
    for k in ("policy", "q1", "q2"):
        named = [(n, tuple(p.shape)) for n, p in nets[k].named_parameters()]
        meta["nets"][k] = [[n, list(s)] for n, s in named]
        vals = fill_params(named, rng)
        with torch.no_grad():
            for n, p in nets[k].named_parameters():
                p.copy_(torch.from_numpy(vals[n]))
    nets["target_q1"].load_state_dict(nets["q1"].state_dict())
    nets["target_q2"].load_state_dict(nets["q2"].state_dict())
    for m in nets.values():
        m.eval()  # dropout off (the reference trains with Dropout(0.1) active: statistically, not bit-, comparable)

    out = {}
    # ---- forward goldens (before any update)
    s, a, r, s2, d = make_batch(rng)
    st, at = torch.from_numpy(s), torch.from_numpy(a)
    with torch.no_grad():
        mean, log_std, value = nets["policy"](st)
        q1 = nets["q1"](torch.cat([st, at], -1)).squeeze(-1)
        q2 = nets["q2"](torch.cat([st, at], -1)).squeeze(-1)
        # F9: the reference indexes the positional encoding by batch row; row 0 == the B=1 result
        mean_row0 = torch.stack([nets["policy"](st[i:i + 1])[0][0] for i in range(8)])
        logstd_row0 = torch.stack([nets["policy"](st[i:i + 1])[1][0] for i in range(8)])
    out["fwd_mean_batchpe"] = mean.numpy()
    out["fwd_logstd_batchpe"] = log_std.numpy()
    out["fwd_mean_pe0_first8"] = mean_row0.numpy()
    out["fwd_logstd_pe0_first8"] = logstd_row0.numpy()
    out["fwd_q1"] = q1.numpy()
    out["fwd_q2"] = q2.numpy()
    # physics-informed loss (reported only, never back-propagated, agent/...:883-906)
    pl, parts = agent.physics_loss(st, at, torch.from_numpy(s2))
    out["physics_loss"] = np.array([float(pl)] + [float(parts[k]) for k in
                                                  ("momentum_conservation", "energy_conservation", "quaternion_normalization")])
    # get_action(deterministic=True) (agent/...:736-809): clamp(mean, -1, 1)
    act_np, info = agent.get_action(st[:16], deterministic=True, algorithm="sac")
    out["get_action_det"] = np.asarray(act_np, dtype=np.float32)

    # ---- three SAC updates with captured noise
    eps_queue = []
    orig_sample, orig_rsample = D.Normal.sample, D.Normal.rsample

    def patched_sample(self, sample_shape=torch.Size()):
        with torch.no_grad():
            return self.loc + self.scale * eps_queue.pop(0)

    def patched_rsample(self, sample_shape=torch.Size()):
        return self.loc + self.scale * eps_queue.pop(0)

    D.Normal.sample, D.Normal.rsample = patched_sample, patched_rsample
    n_updates = 3
    losses = np.zeros((n_updates, 4))
    try:
        for u in range(n_updates):
            # batch-row PE (F9) is part of what the reference computes at B=256; record it as is
            e1 = rng.standard_normal((B, 2)).astype(np.float32)
            e2 = rng.standard_normal((B, 2)).astype(np.float32)
            eps_queue[:] = [torch.from_numpy(e1), torch.from_numpy(e2)]
            if u > 0:
                s, a, r, s2, d = make_batch(rng)
            batch = {"states": torch.from_numpy(s), "actions": torch.from_numpy(a), "rewards": torch.from_numpy(r),
                     "next_states": torch.from_numpy(s2), "dones": torch.from_numpy(d)}
            res = agent.update(batch, algorithm="sac")
            assert "error" not in res, res
            losses[u] = [res["q1_loss"], res["q2_loss"], res["policy_loss"], res.get("physics_loss", 0.0)]
            if u in (0, n_updates - 1):
                for k, m in nets.items():
                    out[f"u{u}_{k}_digest"] = np.stack([tensor_digest(p) for _, p in m.named_parameters()])
                for k in ("optimizer_policy", "optimizer_q1"):
                    st_ = sac[k].state_dict()["state"]
                    out[f"u{u}_{k}_expavg_digest"] = np.stack([tensor_digest(st_[i]["exp_avg"]) if i in st_ else np.zeros(6)
                                                              for i in range(len(sac[k].param_groups[0]["params"]))])
    finally:
        D.Normal.sample, D.Normal.rsample = orig_sample, orig_rsample
    out["losses"] = losses
    np.savez_compressed(os.path.join(OUT, "sac_ref.npz"), **out)
    json.dump(meta, open(os.path.join(OUT, "sac_ref_meta.json"), "w"))
    print("losses (q1, q2, policy, physics):\n", losses)
    print("saved", {k: v.shape for k, v in out.items() if k.startswith(("fwd", "losses", "u0_policy"))})


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("needs /root/reference (build container only)")
    main()
