#!/usr/bin/env python3
"""Golden vectors for the reference's hierarchical acting path, from the reference's own classes:
  HierarchicalAgent.high_level_policy / select_goal / get_action   (agent/multi_algorithm_agent.py:353-417)
  = goal logits from Linear(10,256)-GELU-LN-Linear(256,128)-GELU-LN-Linear(128,4), and the goal-conditioned low-level
    TransformerPolicyNetwork(14, 2, NetworkConfig()) -- NetworkConfig() defaults, i.e. WITH the SqueezeExcitation block.
Weights come from the numpy recipe of gen_sac_golden.fill_params (only names/shapes and outputs are stored); nets in .eval().
Build container only (needs /root/reference).  Output: tests/golden/hier_ref.npz, hier_ref_meta.json"""
import importlib.util
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
SEED = 4242
N = 64


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


def make_states(rng, n=N):
    s = rng.standard_normal((n, 10)).astype(np.float32) * 0.5
    q = rng.standard_normal((n, 4)) * np.array([0.3, 0.3, 0.3, 1.0])
    s[:, :4] = (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(np.float32)
    s[:, 7:10] = rng.uniform(0, 1, (n, 3)).astype(np.float32)
    return s


def main():
    sys.dont_write_bytecode = True
    rec = _load("gen_sac_golden", os.path.join(HERE, "gen_sac_golden.py"))
    sys.path.insert(0, REF)
    from agent.multi_algorithm_agent import HierarchicalAgent

    rng = np.random.default_rng(SEED)
    ha = HierarchicalAgent(10, 2, {})
    ha.high_level_policy.eval()
    ha.low_level_policy.eval()
    meta = {}
    for tag, net in (("high", ha.high_level_policy), ("low", ha.low_level_policy)):
        named = [(n, tuple(p.shape)) for n, p in net.named_parameters()]
        vals = rec.fill_params(named, rng)
        with torch.no_grad():
            for n, p in net.named_parameters():
                p.copy_(torch.from_numpy(vals[n]))
        meta[tag] = [[n, list(s)] for n, s in named]
    s = make_states(rng)
    goal = (np.arange(N) % 4).astype(np.int64)
    out = {}
    with torch.no_grad():
        st = torch.from_numpy(s)
        logits = ha.high_level_policy(st)
        out["logits"] = logits.numpy()
        out["goal_probs"] = torch.softmax(logits, dim=-1).numpy()
        # batch call: the reference indexes its positional-encoding table by batch row
        mean, log_std, value = ha.get_action(st, torch.from_numpy(goal))
        out["mean_batch"], out["log_std_batch"] = mean.numpy(), log_std.numpy()
        # row-by-row calls (B = 1, the only shape the reference's trainer ever uses): PE(0) on every row
        m1, l1 = [], []
        for i in range(16):
            m, l, _ = ha.get_action(st[i:i + 1], torch.from_numpy(goal[i:i + 1]))
            m1.append(m.numpy()[0]); l1.append(l.numpy()[0])
        out["mean_b1"], out["log_std_b1"] = np.array(m1), np.array(l1)
        # select_goal draws with torch.multinomial: only its distribution is comparable
        torch.manual_seed(0)
        draws = torch.stack([ha.select_goal(st) for _ in range(200)])
        out["goal_freq"] = np.stack([(draws == g).float().mean(0).numpy() for g in range(4)], axis=1)
    np.savez_compressed(os.path.join(HERE, "hier_ref.npz"), **out)
    json.dump(meta, open(os.path.join(HERE, "hier_ref_meta.json"), "w"))
    print({k: v.shape for k, v in out.items()})
    print("low-level params:", sum(int(np.prod(s)) for _, s in meta["low"]), "high-level:", sum(int(np.prod(s)) for _, s in meta["high"]))


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("needs /root/reference (build container only)")
    main()
