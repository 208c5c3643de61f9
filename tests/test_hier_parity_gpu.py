"""Hierarchical acting path on the GPU (tvc_ai_amd/hierarchical.py) vs the reference's goldens and the torch oracle."""
import numpy as np
import pytest
import torch

from oracle import sac_torch as st
from tests.test_hier_oracle_golden import hier_setup

pytestmark = pytest.mark.gpu


def _policy(H, P, pe_rows=1, max_rows=4096, train_mode=False):
    from tvc_ai_amd.hierarchical import HierarchicalPolicy
    hp = HierarchicalPolicy(10, 2, device="cuda:0", max_rows=max_rows, seed=0, pe_rows=pe_rows, train_mode=train_mode)
    hp.high.load_state_dict(H)
    hp.low.load_reference_state("policy", P)
    return hp


def test_goal_policy_and_low_level_policy_match_reference_goldens():
    g, H, P, s, goal = hier_setup()
    d = torch.device("cuda:0")
    sd, gd = torch.from_numpy(s).to(d), torch.from_numpy(goal).to(d)
    hp = _policy(H, P, pe_rows=1)
    np.testing.assert_allclose(hp.goal_logits(sd).cpu().numpy(), g["logits"], atol=2e-5)
    mean, ls, _ = hp.get_action(sd[:16], gd[:16])
    np.testing.assert_allclose(mean.cpu().numpy(), g["mean_b1"], atol=1e-4)       # PE(0) on every row == reference at B = 1
    np.testing.assert_allclose(ls.cpu().numpy(), g["log_std_b1"], atol=1e-4)
    hp.close()
    hp = _policy(H, P, pe_rows=64)                                                # the reference's batch-row-indexed table
    mean, ls, _ = hp.get_action(sd, gd)
    np.testing.assert_allclose(mean.cpu().numpy(), g["mean_batch"], atol=1e-4)
    np.testing.assert_allclose(ls.cpu().numpy(), g["log_std_batch"], atol=1e-4)
    hp.close()


@pytest.mark.parametrize("n", [8192, 16384])
def test_goal_draw_and_fused_act_path(n):
    """8192 rows: the fused Linear+LayerNorm acting kernels; 16384 rows: the one-launch row-owner kernel with its
    SqueezeExcitation section (fc1 as one blocked tile, fc2 like the embedding) and the 14-wide [state | goal] input"""
    g, H, P, s, _ = hier_setup()
    d = torch.device("cuda:0")
    hp = _policy(H, P, pe_rows=1, max_rows=n)
    gen = torch.Generator().manual_seed(3)
    big = torch.from_numpy(s).repeat(n // s.shape[0], 1) + 0.01 * torch.randn(n, 10, generator=gen)
    u = torch.rand(n, generator=gen)
    eps = torch.randn(n, 2, generator=gen)
    act, mean, ls, goal = hp.act(big.to(d), eps.to(d), u.to(d), clamp=True)
    with torch.no_grad():
        logits = st.goal_logits(H, big)
        want_goal = st.goal_from_uniform(logits, u)
        cdf = torch.cumsum(torch.softmax(logits, -1), -1)
        margin = (cdf - u.unsqueeze(1)).abs().min(dim=1).values   # rows whose draw sits on a bin edge may differ in fp32
        om, ol = st.hierarchical_act(H, P, big, goal.cpu().long())
    got = goal.cpu().long()
    assert ((got == want_goal) | (margin < 1e-5)).all()
    assert (got != want_goal).sum() <= 2
    freq = torch.bincount(got, minlength=4).float() / n
    np.testing.assert_allclose(freq.numpy(), torch.softmax(logits, -1).mean(0).numpy(), atol=0.03)
    np.testing.assert_allclose(mean.cpu().numpy(), om.numpy(), atol=2e-4)
    np.testing.assert_allclose(ls.cpu().numpy(), ol.numpy(), atol=2e-4)
    want_act = torch.clamp(om + torch.exp(ol) * eps, -1, 1)
    np.testing.assert_allclose(act.cpu().numpy(), want_act.numpy(), atol=1e-3)
    hp.close()


def test_low_level_policy_in_train_mode_like_the_reference():
    """the reference's hierarchical nets are never put in eval mode either: the goal-conditioned TransformerPolicyNetwork acts with
    Dropout(0.1) active.  Same masks on both sides (DropMasks, site base 300, call counter), SE block included."""
    g, H, P, s, goal = hier_setup()
    d = torch.device("cuda:0")
    hp = _policy(H, P, pe_rows=1, train_mode=True)
    sd, gd = torch.from_numpy(s).to(d), torch.from_numpy(goal).to(d)
    onehot = torch.nn.functional.one_hot(torch.from_numpy(goal).long(), 4).float()
    sg = torch.cat([torch.from_numpy(s), onehot], -1)
    Pt = {k: torch.as_tensor(v) for k, v in P.items()}
    masks = st.DropMasks(0.1, seed=int(hp.low.cfg.dropout_seed))
    outs = []
    for call in range(2):
        mean, ls, _ = hp.get_action(sd, gd)
        with torch.no_grad():
            m_ref, ls_ref = st.actor_forward(Pt, sg, False, drop=masks.hook(call, 300))
        np.testing.assert_allclose(mean.cpu().numpy(), m_ref.numpy(), atol=2e-4)
        np.testing.assert_allclose(ls.cpu().numpy(), ls_ref.numpy(), atol=2e-4)
        outs.append(mean.cpu())
    assert (outs[0] - outs[1]).abs().max().item() > 1e-3
    hp.close()


def test_agent_acts_through_the_hierarchy_when_enabled():
    from tvc_ai_amd.agent import MultiAlgorithmAgent
    cfg = {"hierarchical_rl": {"enabled": True}, "tvc_native": {"batch_size": 1, "max_act_rows": 64}}
    ag = MultiAlgorithmAgent(10, 2, cfg, device="cuda:0")
    s = torch.randn(5, 10)
    a, info = ag.get_action(s)
    assert a.shape == (5, 2) and np.all(np.abs(a) <= 1.0) and info["goal"].shape == (5,) and set(info["goal"]) <= {0, 1, 2, 3}
    a2, info2 = ag.get_action(s, deterministic=True)
    assert np.isfinite(a2).all() and info2["mean"].shape == (5, 2)
    flat = MultiAlgorithmAgent(10, 2, {"tvc_native": {"batch_size": 1, "max_act_rows": 64}}, device="cuda:0")
    assert flat.hierarchical_agent is None and "goal" not in flat.get_action(s)[1]
