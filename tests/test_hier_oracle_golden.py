"""Torch restatement of the reference's hierarchical acting path (oracle/sac_torch.py: goal_logits, hierarchical_act) vs goldens
produced by the reference's own HierarchicalAgent (tests/golden/gen_hier_golden.py)."""
import importlib.util
import json
import os

import numpy as np
import torch

from oracle import sac_torch as st

HERE = os.path.dirname(os.path.abspath(__file__))


def _mod(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(HERE, "golden", name + ".py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def hier_setup():
    """-> goldens, high-level params, low-level params, states, goals (same numpy recipe as the generator)"""
    rec, gen = _mod("gen_sac_golden"), _mod("gen_hier_golden")
    g = np.load(os.path.join(HERE, "golden", "hier_ref.npz"))
    meta = json.load(open(os.path.join(HERE, "golden", "hier_ref_meta.json")))
    rng = np.random.default_rng(gen.SEED)
    H = {k: torch.from_numpy(v) for k, v in rec.fill_params([(n, tuple(s)) for n, s in meta["high"]], rng).items()}
    P = {k: torch.from_numpy(v) for k, v in rec.fill_params([(n, tuple(s)) for n, s in meta["low"]], rng).items()}
    s = gen.make_states(rng)
    goal = (np.arange(gen.N) % 4).astype(np.int64)
    return g, H, P, s, goal


def test_goal_policy_and_low_level_policy_match_reference():
    g, H, P, s, goal = hier_setup()
    with torch.no_grad():
        st_ = torch.from_numpy(s)
        logits = st.goal_logits(H, st_)
        mb, lb = st.hierarchical_act(H, P, st_, torch.from_numpy(goal), batch_pe=True)
        m1, l1 = st.hierarchical_act(H, P, st_[:16], torch.from_numpy(goal[:16]), batch_pe=False)
    np.testing.assert_allclose(logits.numpy(), g["logits"], atol=3e-6)
    np.testing.assert_allclose(mb.numpy(), g["mean_batch"], atol=2e-5)
    np.testing.assert_allclose(lb.numpy(), g["log_std_batch"], atol=2e-5)
    np.testing.assert_allclose(m1.numpy(), g["mean_b1"], atol=2e-5)
    np.testing.assert_allclose(l1.numpy(), g["log_std_b1"], atol=2e-5)


def test_goal_draw_follows_the_softmax():
    g, H, _, s, _ = hier_setup()
    with torch.no_grad():
        logits = st.goal_logits(H, torch.from_numpy(s))
        gen = torch.Generator().manual_seed(1)
        draws = torch.stack([st.goal_from_uniform(logits, torch.rand(logits.shape[0], generator=gen)) for _ in range(2000)])
    freq = np.stack([(draws == k).float().mean(0).numpy() for k in range(4)], axis=1)
    np.testing.assert_allclose(freq, g["goal_probs"], atol=0.05)           # inverse-CDF draw vs the reference's probabilities
    np.testing.assert_allclose(g["goal_freq"], g["goal_probs"], atol=0.12)  # the reference's own 200 multinomial draws agree too
