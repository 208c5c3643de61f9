"""Split-operand acting kernel (tvc_actor_x3.h, tvc_sac_act flags bit 4): every Linear of the policy on the bf16 matrix pipe with both
operands written as three bf16 terms, six products, fp32 accumulate.

What is checked: (1) against the fp64 evaluation of the restatement (oracle/sac_torch.actor_forward on float64 tensors) the kernel is
as close as the exact f32-MFMA kernel -- max error <= 1.5 x the f32 kernel's + 2e-6, and within the 3e-4 bar every acting test
uses; (2) the snapshot / share-CUs forms are bit-equal to the plain call; (3) a policy update re-packs the stream."""
import numpy as np
import pytest
import torch

from oracle import sac_torch as st

pytestmark = pytest.mark.gpu


def cuda(*xs):
    return [torch.as_tensor(x).cuda().contiguous() for x in xs]


def _perturb(sac):
    for name, _, rows, cols in sac.table:  # non-trivial norms / biases so that every epilogue term is exercised
        if name.startswith("policy.") and cols == 1:
            sac.view(name).add_(0.1 * torch.randn(rows, device="cuda", generator=torch.Generator(device="cuda").manual_seed(rows)))
    sac.sync_derived()


def _ref64(P, obs):
    P64 = {k: v.double() for k, v in P.items()}
    with torch.no_grad():
        return st.actor_forward(P64, obs.double(), batch_pe=False)


@pytest.mark.parametrize("n,use_se", [(65536, 0), (65536 - 37, 0), (40000 + 21, 0), (40005, 1)])
def test_x3_acting_is_as_close_to_fp64_as_the_f32_kernel(n, use_se):
    from tvc_ai_amd.agent import NativeSAC, sac_cfg
    torch.set_num_threads(8)
    obs_dim = 14 if use_se else 10
    sac = NativeSAC(sac_cfg(0, obs_dim=obs_dim, batch_size=64, max_act_rows=65536, use_se=use_se), seed=13)
    _perturb(sac)
    P = sac.export_reference_state("policy")
    if use_se:  # (the reference-keyed export covers the SAC policy's tensors; the SE block belongs to the hierarchical policy)
        for key in ("se_block.fc1.weight", "se_block.fc1.bias", "se_block.fc2.weight", "se_block.fc2.bias"):
            P[key] = sac.view("policy." + key).detach().cpu().clone()
    g = torch.Generator().manual_seed(n)
    obs = torch.randn(n, obs_dim, generator=g) * 0.5
    eps = torch.randn(n, 2, generator=g)
    o, e = cuda(obs, eps)
    a32, m32, l32 = [t.clone() for t in sac.act(o, e)]
    a3, m3, l3 = [t.clone() for t in sac.act(o, e, x3=True)]
    assert torch.isfinite(m3).all() and torch.isfinite(l3).all()
    pick = torch.cat([torch.arange(0, 300), torch.randint(0, n, (1200,), generator=g), torch.arange(n - 300, n)])
    m_ref, ls_ref = _ref64(P, obs[pick])
    def err(m, ls):
        return max(float((m.cpu()[pick].double() - m_ref).abs().max()), float((ls.cpu()[pick].double() - ls_ref).abs().max()))
    e32, e3 = err(m32, l32), err(m3, l3)
    from tests import parity_log
    parity_log.record(f"acting_x3_vs_fp64_n{n}_se{use_se}", max_err_x3=e3, max_err_f32_mfma=e32,
                      max_abs_diff_x3_f32=float((m3 - m32).abs().max()))
    assert e3 <= 3e-4, e3
    assert e3 <= 1.5 * e32 + 2e-6, (e3, e32)
    a_ref = (m_ref + torch.exp(ls_ref) * eps[pick].double()).clamp(-1, 1)
    np.testing.assert_allclose(a3.cpu()[pick].double().numpy(), a_ref.numpy(), atol=1e-3, rtol=0)
    # all rows against the f32 kernel (both are fp32-exact to rounding: a few 1e-6)
    assert float((m3 - m32).abs().max()) <= 5e-5 and float((l3 - l32).abs().max()) <= 5e-5
    # snapshot + share-CUs forms: same kernel, same tile stream -> bit-equal
    sac.snapshot_policy()
    k = (n // 2) // 64 * 64
    outs = tuple(torch.empty(n, 2, device="cuda") for _ in range(3))
    sac.act(o[:k], e[:k], out=tuple(t[:k] for t in outs), snapshot=True, share_cus=True, x3=True)
    sac.act(o[k:], e[k:], out=tuple(t[k:] for t in outs), snapshot=True, x3=True)
    assert torch.equal(outs[0], a3) and torch.equal(outs[1], m3) and torch.equal(outs[2], l3)
    sac.close()


def test_x3_stream_is_repacked_by_a_policy_update():
    from tvc_ai_amd.agent import NativeSAC, sac_cfg
    n, B = 16384, 64
    sac = NativeSAC(sac_cfg(0, batch_size=B, max_act_rows=n, dropout_p=0.0), seed=5)
    g = torch.Generator().manual_seed(1)
    obs = torch.randn(n, 10, generator=g) * 0.5
    o, = cuda(obs)
    sac.enable_x3()
    m_before = sac.act(o, None, x3=True)[1].clone()
    s, s2 = torch.randn(B, 10, generator=g), torch.randn(B, 10, generator=g)
    a, r, d = torch.rand(B, 2, generator=g) * 2 - 1, torch.randn(B, generator=g), torch.zeros(B)
    e1, e2 = torch.randn(B, 2, generator=g), torch.randn(B, 2, generator=g)
    for _ in range(3):
        sac.update(*cuda(s, a, r, s2, d, e1, e2))
    P = sac.export_reference_state("policy")
    m3 = sac.act(o, None, x3=True)[1]
    assert float((m3 - m_before).abs().max()) > 1e-4  # the policy moved
    pick = torch.arange(0, n, 37)
    m_ref, _ = _ref64(P, obs[pick])
    assert float((m3.cpu()[pick].double() - m_ref).abs().max()) <= 3e-4
    sac.close()


def test_train_loop_with_x3_acting_matches_the_f32_loop_on_its_first_step_and_stays_finite():
    """VecTrainer(acting_x3=True): snapshot + split (share-CUs / exclusive) forms of actor_x3_kernel in the two-stream loop.  The first
    step's actions equal the f32 loop's to rounding (same seeds, same policy); after that the trajectories may part at threshold
    terms, so the rest is a finiteness / progress check."""
    from tvc_ai_amd.trainer import VecTrainer
    n = 32768
    acts, tr = [], None
    for x3 in (False, True):
        tr = VecTrainer(n, family=0, batch_size=64, replay_capacity=1 << 17, seed=3, acting_x3=x3, defer_join=True)
        torch.manual_seed(7)  # (the acting noise comes from torch's global generator)
        tr.step(True)
        torch.cuda.synchronize()
        acts.append(tr.act.clone())
        if not x3:
            tr.close()
    d = float((acts[0] - acts[1]).abs().max())
    assert d <= 5e-5, d
    for _ in range(6):
        tr.step(True)
    torch.cuda.synchronize()
    st = tr.stats()
    assert all(np.isfinite(st["losses"])), st
    assert torch.isfinite(tr.act).all()
    from tests import parity_log
    parity_log.record("train_loop_x3_first_step_action_diff", max_abs_diff=d, envs=n)
    tr.close()


def _randomise_vectors(sac, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    for name, _, rows, cols in sac.table:
        if name.startswith("policy.") and cols == 1:
            sac.view(name).add_(0.1 * torch.randn(rows, device="cuda", generator=g))
    sac.sync_derived()


def test_train_mode_acting_with_split_operands_matches_the_restatement_mask_for_mask():
    """actor_x3_kernel<true> (tvc_sac_act flags bits 3 + 4, >= 16 384 rows): the net as trained with every Dropout live, on the bf16 matrix
    pipe with split operands.  Against the eager restatement with the kernels' own hash masks (DropMasks, site base 300, counter =
    acting calls so far) element for element -- live parameters, snapshot, CU-sharing form -- and within rounding of the f32
    train-mode kernel (actor_split_kernel<true>) at the same call index of an identical handle."""
    from tvc_ai_amd.agent import NativeSAC, sac_cfg
    from tests import parity_log
    torch.set_num_threads(8)
    n, p = 20000, 0.1
    sacs = [NativeSAC(sac_cfg(0, batch_size=64, max_act_rows=32768, dropout_p=p), seed=23) for _ in range(2)]
    for s_ in sacs:
        _randomise_vectors(s_, 9)
    sac, ref32 = sacs
    P = sac.export_reference_state("policy")
    g = torch.Generator().manual_seed(n)
    obs = torch.randn(n, 10, generator=g) * 0.5
    eps = torch.randn(n, 2, generator=g)
    og, eg = obs.cuda(), eps.cuda()
    masks = st.DropMasks(p, seed=int(sac.cfg.dropout_seed))
    pick = torch.cat([torch.arange(0, 64), torch.randint(0, n, (400,), generator=g), torch.arange(n - 64, n)])
    worst = worst32 = 0.0
    sac.snapshot_policy()
    for call, kw in enumerate((dict(), dict(snapshot=True), dict(snapshot=True, share_cus=True))):
        act, mean, ls = sac.act(og, eg, train_mode=True, x3=True, **kw)
        assert sac.act_counter() == call + 1
        _, mean32, ls32 = ref32.act(og, eg, train_mode=True)  # same weights, same seed, same call index: the same masks
        worst32 = max(worst32, float((mean - mean32).abs().max()), float((ls - ls32).abs().max()))
        with torch.no_grad():
            m_ref, ls_ref = st.actor_forward(P, obs, False, drop=masks.hook(call, 300))
        a_ref = torch.clamp(m_ref + torch.exp(ls_ref) * eps, -1, 1)
        for got, want in ((mean, m_ref), (ls, ls_ref), (act, a_ref)):
            err = (got.cpu()[pick] - want[pick]).abs().max().item()
            worst = max(worst, err)
            assert err <= 3e-4 * max(1.0, want.abs().max().item()), (call, kw, err)
    assert worst32 <= 1e-4, worst32
    parity_log.record("train_mode_acting_x3_n20000", rows=n, calls=3, worst_abs_err=worst, worst_vs_f32_train_kernel=worst32)
    for s_ in sacs:
        s_.close()


def test_segment_graphs_replay_the_loop_with_split_operand_acting():
    """the launch mode of a multi-rank job (4 graph launches per step around the update's collectives) with acting_x3: the captured
    packing launch carries the split-operand stream, the captured acting launches read its snapshot"""
    from tvc_ai_amd.trainer import VecTrainer
    tr = VecTrainer(32768, family=0, batch_size=64, replay_capacity=1 << 17, seed=5, acting_x3=True, defer_join=True)
    for _ in range(3):
        tr.step(True)
    seg = tr.capture_segments()
    p0 = tr.sac.params.clone()
    for _ in range(6):
        seg()
    torch.cuda.synchronize()
    assert torch.isfinite(tr.sac.params).all() and torch.isfinite(tr.act).all()
    assert float((tr.sac.params - p0).abs().max()) > 0.0  # the replayed updates step the parameters
    # the replayed acting launches follow the policy: a live x3 call on the current observations equals the step's own actions' means
    obs = tr.obs[tr.cur].clone()
    m_live = tr.sac.act(obs, None, x3=True)[1]
    assert torch.isfinite(m_live).all()
    tr.close()


def test_hierarchical_acting_path_takes_the_split_operand_kernel():
    """HierarchicalPolicy.act(x3=True): the goal-conditioned low-level policy (SqueezeExcitation block, 14-wide input) through
    actor_x3_kernel -- same goals, actions within rounding of the f32 kernel's, also with the rows split into a CU-sharing and an
    exclusive launch"""
    from tvc_ai_amd.hierarchical import HierarchicalPolicy
    n = 40000
    hp = HierarchicalPolicy(10, 2, device="cuda:0", max_rows=n, seed=3)
    g = torch.Generator(device="cuda").manual_seed(2)
    ob = torch.randn(n, 10, device="cuda", generator=g) * 0.5
    ep, uu = torch.randn(n, 2, device="cuda", generator=g), torch.rand(n, device="cuda", generator=g)
    a0, m0, l0, g0 = [t.clone() for t in hp.act(ob, ep, uu)]
    a1, m1, l1, g1 = [t.clone() for t in hp.act(ob, ep, uu, x3=True)]
    assert torch.equal(g0, g1)
    d = max(float((m0 - m1).abs().max()), float((l0 - l1).abs().max()))
    assert d <= 5e-5, d
    a2, m2, l2, g2 = hp.act(ob, ep, uu, x3=True, share_rows=20000)
    assert torch.equal(m2, m1) and torch.equal(a2, a1)
    from tests import parity_log
    parity_log.record("hierarchical_acting_x3", rows=n, max_abs_diff_vs_f32=d)
    hp.close()
