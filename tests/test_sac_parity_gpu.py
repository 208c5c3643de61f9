"""HIP SAC learner (through the C ABI) vs the eager-PyTorch restatement (oracle/sac_torch.py, itself pinned
against the reference's own code) and vs the reference goldens (data only).

Tolerances (fp32 on both sides; MFMA f32 is an exact k-ordered fma chain, so differences are summation order):
  forward outputs |diff| <= 2e-4 abs, losses <= 2e-4 rel, parameters after k Adam steps <= 2*lr*k*1.05 per element
  (an element whose gradient is ~0 moves by lr per step in either direction), < 2 % of elements off by > 2e-5, and <= 2e-5 relative on the
  per-tensor |.|-sum digest."""
import importlib.util
import json
import os

import numpy as np
import pytest
import torch

from oracle import sac_torch as st

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _recipe():
    spec = importlib.util.spec_from_file_location("gen_sac_golden", os.path.join(HERE, "golden", "gen_sac_golden.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def ref_nets(rec, meta, rng):
    nets = {}
    for k in ("policy", "q1", "q2"):
        named = [(n, tuple(s)) for n, s in meta["nets"][k]]
        nets[k] = {n: torch.from_numpy(v) for n, v in rec.fill_params(named, rng).items()}
    return nets


def load_into_native(sac, nets):
    sac.load_reference_state("policy", nets["policy"])
    for k in ("q1", "q2"):
        sac.load_reference_state(k, nets[k])
        sac.load_reference_state("target_" + k, nets[k])


def cuda(*xs):
    return [torch.as_tensor(x).cuda().contiguous() for x in xs]


@pytest.mark.parametrize("batch_pe", [True, False])
def test_reference_shapes_forward_and_three_updates(batch_pe):
    from tvc_ai_amd.agent import NativeSAC, sac_cfg
    torch.set_num_threads(8)
    rec = _recipe()
    meta = json.load(open(os.path.join(HERE, "golden", "sac_ref_meta.json")))
    g = np.load(os.path.join(HERE, "golden", "sac_ref.npz"))
    rng = np.random.default_rng(meta["seed"])
    nets = ref_nets(rec, meta, rng)
    B = 256
    sac = NativeSAC(sac_cfg(0, batch_size=B, max_act_rows=1024, pe_rows=B if batch_pe else 1), init=False)
    load_into_native(sac, nets)
    s, a, r, s2, d = [torch.from_numpy(x) for x in rec.make_batch(rng)]
    sg, ag, rg, s2g, dg = cuda(s, a, r, s2, d)

    # forward: actor (deterministic act) and critics
    act, mean, ls = sac.act(sg, None)
    with torch.no_grad():
        m_ref, ls_ref = st.actor_forward(nets["policy"], s, batch_pe=batch_pe)
        q_ref = torch.stack([st.critic_forward(nets["q1"], s, a), st.critic_forward(nets["q2"], s, a)])
    np.testing.assert_allclose(mean.cpu().numpy(), m_ref.numpy(), atol=2e-4, rtol=0)
    np.testing.assert_allclose(ls.cpu().numpy(), ls_ref.numpy(), atol=2e-4, rtol=0)
    np.testing.assert_allclose(act.cpu().numpy(), m_ref.clamp(-1, 1).numpy(), atol=2e-4, rtol=0)
    q = sac.q_values(sg, ag)
    np.testing.assert_allclose(q.cpu().numpy(), q_ref.numpy(), atol=2e-4, rtol=2e-5)
    if batch_pe:  # the reference's own numbers (batch-row positional encoding)
        np.testing.assert_allclose(mean.cpu().numpy(), g["fwd_mean_batchpe"], atol=2e-4, rtol=0)
        np.testing.assert_allclose(q[0].cpu().numpy(), g["fwd_q1"], atol=2e-4, rtol=2e-5)
    else:
        np.testing.assert_allclose(mean[:1].cpu().numpy(), g["fwd_mean_pe0_first8"][:1], atol=2e-4, rtol=0)

    # three updates with the captured noise
    orc = st.SacOracle(nets["policy"], nets["q1"], nets["q2"], batch_pe=batch_pe)
    for u in range(3):
        e1 = torch.from_numpy(rng.standard_normal((B, 2)).astype(np.float32))
        e2 = torch.from_numpy(rng.standard_normal((B, 2)).astype(np.float32))
        if u > 0:
            s, a, r, s2, d = [torch.from_numpy(x) for x in rec.make_batch(rng)]
            sg, ag, rg, s2g, dg = cuda(s, a, r, s2, d)
        l_ref = orc.update(s, a, r, s2, d, e1, e2)
        pl_ref, _ = st.physics_loss(s, a, s2)
        losses = sac.update(sg, ag, rg, s2g, dg, *cuda(e1, e2)).cpu().numpy()
        np.testing.assert_allclose(losses[:3], l_ref, rtol=2e-4, err_msg=f"update {u}")
        np.testing.assert_allclose(losses[3], float(pl_ref), rtol=1e-4)
        if batch_pe:
            np.testing.assert_allclose(losses[:3], g["losses"][u, :3], rtol=2e-4, err_msg=f"update {u} vs reference golden")
        # parameters
        # Adam's first steps are sign-like (lr * g / (|g| + eps)): an element whose gradient is ~0 can come out
        # at +lr on one side and -lr on the other, so single elements may differ by 2 lr per step; almost all
        # elements must agree far better than that
        tol = 2 * st.LR * (u + 1) * 1.05
        groups = {"policy": orc.P, "q1": orc.Q[0], "q2": orc.Q[1], "target_q1": orc.TQ[0], "target_q2": orc.TQ[1]}
        for net, ref_params in groups.items():
            mine = sac.export_reference_state(net)
            for k, v in ref_params.items():
                if net == "policy" and (k.startswith("value_head") or k.startswith("pos_encoding")):
                    continue
                got, want = mine[k], v.detach()
                if "in_proj" in k:
                    got, want = got[512:], want[512:]
                adiff = (got - want).abs()
                diff = adiff.max().item()
                assert diff <= tol, (u, net, k, diff)
                assert (adiff > 2e-5).float().mean().item() < 0.02, (u, net, k, "too many elements off")
                sa, sb = got.abs().sum().item(), want.abs().sum().item()
                assert abs(sa - sb) <= 2e-5 * max(sb, 1.0) + 1e-6, (u, net, k, sa, sb)
    sac.close()


def test_mlp_family_updates():
    """BASELINE.json's 256x256 ReLU MLP actor/critics (legacy SACAgent shapes), same update rule."""
    from tvc_ai_amd.agent import NativeSAC, sac_cfg
    torch.manual_seed(3)
    B = 256
    sac = NativeSAC(sac_cfg(1, batch_size=B, max_act_rows=512), seed=5)
    P = {k[len("policy."):]: sac.view(k).cpu().clone() for k, *_ in sac.table if k.startswith("policy.")}
    Q1 = {k[len("q1."):]: sac.view(k).cpu().clone() for k, *_ in sac.table if k.startswith("q1.")}
    Q2 = {k[len("q2."):]: sac.view(k).cpu().clone() for k, *_ in sac.table if k.startswith("q2.")}
    orc = st.SacOracle(P, Q1, Q2, actor_fn=st.mlp_actor_forward, critic_fn=st.mlp_critic_forward)
    for u in range(4):
        s = torch.randn(B, 10) * 0.5
        s2 = s + 0.05 * torch.randn(B, 10)
        a = torch.rand(B, 2) * 2 - 1
        r = torch.randn(B) * 30 + 50
        d = (torch.rand(B) < 0.1).float()
        e1, e2 = torch.randn(B, 2), torch.randn(B, 2)
        l_ref = orc.update(s, a, r, s2, d, e1, e2)
        losses = sac.update(*cuda(s, a, r, s2, d, e1, e2)).cpu().numpy()
        np.testing.assert_allclose(losses[:3], l_ref, rtol=3e-4, err_msg=f"update {u}")
    tol = 2 * st.LR * 4 * 1.05
    for name, ref in (("policy.", orc.P), ("q1.", orc.Q[0]), ("target_q2.", orc.TQ[1])):
        for k, v in ref.items():
            assert (sac.view(name + k).cpu() - v.detach()).abs().max().item() <= tol, (name, k)
    sac.close()


@pytest.mark.parametrize("n", [700, 6144 + 77])
def test_mlp_family_acting(n):
    """acting pass of the MLP family below and above the row count where the hidden layer and the head share one launch"""
    from tvc_ai_amd.agent import NativeSAC, sac_cfg
    sac = NativeSAC(sac_cfg(1, batch_size=64, max_act_rows=8192), seed=17)
    P = {k[len("policy."):]: sac.view(k).cpu().clone() for k, *_ in sac.table if k.startswith("policy.")}
    obs = torch.randn(n, 10) * 0.5
    eps = torch.randn(n, 2)
    act, mean, ls = sac.act(*cuda(obs), cuda(eps)[0])
    with torch.no_grad():
        m_ref, ls_ref = st.mlp_actor_forward(P, obs)
    np.testing.assert_allclose(mean.cpu().numpy(), m_ref.numpy(), atol=1e-4, rtol=0)
    np.testing.assert_allclose(ls.cpu().numpy(), ls_ref.numpy(), atol=1e-4, rtol=0)
    np.testing.assert_allclose(act.cpu().numpy(), (m_ref + torch.exp(ls_ref) * eps).clamp(-1, 1).numpy(), atol=5e-4, rtol=0)
    sac.close()


def test_act_large_ragged_batch_and_sampling():
    from tvc_ai_amd.agent import NativeSAC, sac_cfg
    n = 5000 + 37
    sac = NativeSAC(sac_cfg(0, batch_size=64, max_act_rows=8192), seed=11)
    P = sac.export_reference_state("policy")
    obs = torch.randn(n, 10) * 0.5
    eps = torch.randn(n, 2)
    act, mean, ls = sac.act(*cuda(obs), cuda(eps)[0])
    with torch.no_grad():
        m_ref, ls_ref = st.actor_forward(P, obs, batch_pe=False)
    np.testing.assert_allclose(mean.cpu().numpy(), m_ref.numpy(), atol=3e-4, rtol=0)
    np.testing.assert_allclose(ls.cpu().numpy(), ls_ref.numpy(), atol=3e-4, rtol=0)
    a_ref = (m_ref + torch.exp(ls_ref) * eps).clamp(-1, 1)
    np.testing.assert_allclose(act.cpu().numpy(), a_ref.numpy(), atol=1e-3, rtol=0)
    with pytest.raises(Exception):
        sac.act(torch.zeros(9000, 10).cuda(), None)  # > max_act_rows must fail loudly
    sac.close()


@pytest.mark.parametrize("n", [65536, 65536 - 37, 6144 + 5])
def test_act_at_full_chip_row_counts(n):
    """the acting kernels that only run at large row counts: 128x128-per-wave tiles (one workgroup per CU, n >= ~56k) and the
    32-row fused Linear+LayerNorm kernel (n >= 6144), checked on a sample of rows against the restatement (PE(0): rows are
    independent), ragged last tile included"""
    from tvc_ai_amd.agent import NativeSAC, sac_cfg
    torch.set_num_threads(8)
    sac = NativeSAC(sac_cfg(0, batch_size=64, max_act_rows=65536), seed=13)
    for name, _, rows, cols in sac.table:  # non-trivial norms / biases so that every epilogue term is exercised
        if name.startswith("policy.") and cols == 1:
            sac.view(name).add_(0.1 * torch.randn(rows, device="cuda", generator=torch.Generator(device="cuda").manual_seed(rows)))
    sac.sync_derived()
    P = sac.export_reference_state("policy")
    g = torch.Generator().manual_seed(n)
    obs = torch.randn(n, 10, generator=g) * 0.5
    eps = torch.randn(n, 2, generator=g)
    act, mean, ls = sac.act(*cuda(obs), cuda(eps)[0])
    pick = torch.cat([torch.arange(0, 300), torch.randint(0, n, (1200,), generator=g), torch.arange(n - 300, n)])
    with torch.no_grad():
        m_ref, ls_ref = st.actor_forward(P, obs[pick], batch_pe=False)
    np.testing.assert_allclose(mean.cpu()[pick].numpy(), m_ref.numpy(), atol=3e-4, rtol=0)
    np.testing.assert_allclose(ls.cpu()[pick].numpy(), ls_ref.numpy(), atol=3e-4, rtol=0)
    a_ref = (m_ref + torch.exp(ls_ref) * eps[pick]).clamp(-1, 1)
    np.testing.assert_allclose(act.cpu()[pick].numpy(), a_ref.numpy(), atol=1e-3, rtol=0)
    assert torch.isfinite(mean).all() and torch.isfinite(ls).all()
    sac.close()


def test_sampling_a_fresh_replay_buffer_returns_zero_rows_not_garbage():
    """the buffer is zero-filled at create, so the legacy store_transition / sample surface cannot feed NaNs."""
    from tvc_ai_amd.agent import ReplayBuffer
    rb = ReplayBuffer(50_000, 10, 2, seed=1)
    out = rb.sample(256)
    for t in out:
        assert torch.isfinite(t).all() and (t == 0).all()
    rb.close()


def test_replay_buffer_roundtrip_and_uniformity():
    from tvc_ai_amd.agent import ReplayBuffer
    rb = ReplayBuffer(1000, 10, 2, seed=7)
    assert len(rb) == 0
    n = 300
    rows = []
    for k in range(5):  # 1500 rows into capacity 1000: ring overwrite
        s = torch.full((n, 10), float(k)).cuda() + torch.arange(n).view(n, 1).cuda() * 1e-3
        a = torch.rand(n, 2).cuda()
        r = torch.arange(n).float().cuda() + 1000 * k
        s2 = s + 0.5
        term = (torch.arange(n) % 7 == 0).to(torch.uint8).cuda()
        trunc = (torch.arange(n) % 11 == 0).to(torch.uint8).cuda()
        rb.insert(s, a, r, s2, term, trunc)
        rows.append((s, a, r, s2, term | trunc))
    assert len(rb) == 1000
    s, a, r, s2, d = rb.sample(4096, counter=3)
    r_np = r.cpu().numpy()
    assert r_np.min() >= 1000 * 1 + 200  # rows 0..499 (k=0 and most of k=1) were overwritten
    # each sampled row is internally consistent
    k = (r_np // 1000).astype(int)
    i = (r_np % 1000).astype(int)
    np.testing.assert_allclose(s[:, 0].cpu().numpy(), k + i * 1e-3, atol=1e-5)
    np.testing.assert_allclose(s2.cpu().numpy(), s.cpu().numpy() + 0.5, atol=1e-6)
    np.testing.assert_array_equal(d.cpu().numpy(), ((i % 7 == 0) | (i % 11 == 0)).astype(np.float32))
    # same (seed, counter) -> same rows; device counter advances
    s_again, *_ = rb.sample(4096, counter=3)
    assert torch.equal(s, s_again)
    x1 = rb.sample(256)[2].clone()
    x2 = rb.sample(256)[2].clone()
    assert not torch.equal(x1, x2)
    # roughly uniform over the 1000 live rows
    _, _, rr, _, _ = rb.sample(200000, counter=9)
    cnt = np.bincount(((rr.cpu().numpy() // 1000) * 300 + rr.cpu().numpy() % 1000).astype(int))
    live = cnt[cnt > 0]
    assert len(live) == 1000 and live.min() > 120 and live.max() < 290
    rb.close()


def test_agent_surface_update_accepts_bool_dones():
    """MultiAlgorithmAgent drop-in: scripts/train.py:577-584 builds B=1 tensors with a BoolTensor `dones`."""
    from tvc_ai_amd.agent import MultiAlgorithmAgent
    agent = MultiAlgorithmAgent(10, 2, {"tvc_native": {"batch_size": 1, "max_act_rows": 16}, "physics_informed": {"enabled": True}})
    assert list(agent.algorithms) == ["ppo", "sac", "td3"]  # the reference builds all three by default (:487-497)
    obs = torch.randn(1, 10)
    action, info = agent.get_action(obs, algorithm="sac")
    assert isinstance(action, np.ndarray) and action.shape == (1, 2) and np.all(np.abs(action) <= 1)
    assert info["algorithm"] == "sac" and info["mean"].shape == (1, 2)
    batch = {"states": obs, "actions": torch.from_numpy(action), "rewards": torch.tensor([1.5]),
             "next_states": obs + 0.1, "dones": torch.BoolTensor([False])}
    out = agent.update(batch, algorithm="sac")
    assert set(out) >= {"q1_loss", "q2_loss", "policy_loss"} and all(np.isfinite(v) for v in out.values()), out
    agent.update_performance("sac", 12.0)
    assert list(agent.performance_history["sac"]) == [12.0]
    assert agent.select_algorithm() == "sac"  # the only algorithm with a history (:700-706)


def test_select_algorithm_and_the_eager_passthrough_follow_the_reference():
    """agent/...:693-709, 736-809, 868-948, 1018-1086: with no history the 'dynamic' strategy answers 'ppo' (so the reference's
    own loop, scripts/train.py:544-584, acts and updates through PPO); PPO / TD3 are the eager-PyTorch pass-through here; the
    'voting' strategy acts with the weighted ensemble of all three policies, the SAC one through the HIP kernels."""
    from tvc_ai_amd.agent import MultiAlgorithmAgent
    cfg = {"tvc_native": {"batch_size": 8, "max_act_rows": 16}}
    agent = MultiAlgorithmAgent(10, 2, cfg, seed=3)
    assert agent.select_algorithm() == "ppo"
    obs = torch.randn(8, 10)
    act, info = agent.get_action(obs)
    assert info["algorithm"] == "ppo" and act.shape == (8, 2) and np.all(np.abs(act) <= 1)
    batch = {"states": obs, "actions": torch.from_numpy(act), "rewards": torch.randn(8), "next_states": obs + 0.1,
             "dones": torch.BoolTensor([False] * 8)}
    p0 = [p.detach().clone() for p in agent.algorithms["ppo"]["policy"].parameters()]
    out = agent.update(batch)  # algorithm None -> select_algorithm() -> 'ppo'
    assert set(out) == {"policy_loss", "value_loss", "total_loss"} and all(np.isfinite(v) for v in out.values())
    assert any(not torch.equal(a, b) for a, b in zip(p0, agent.algorithms["ppo"]["policy"].parameters()))
    t1 = agent.update(batch, algorithm="td3")
    t2 = agent.update(batch, algorithm="td3")
    assert t1["policy_loss"] == 0.0 and t2["policy_loss"] != 0.0  # delayed policy update: every second call (:1060-1066)
    a3, i3 = agent.get_action(obs, algorithm="td3", deterministic=True)
    assert np.all(np.abs(a3) <= 1) and np.all(i3["log_std"] == 0)
    for _ in range(3):
        agent.update_performance("td3", 5.0)
    agent.update_performance("sac", 9.0)
    assert agent.select_algorithm() == "sac"
    agent.selection_strategy = "voting"
    ae, ie = agent.get_action(obs)
    assert ie["algorithm"] == "ensemble" and len(ie["individual_actions"]) == 3 and abs(float(ie["weights"].sum()) - 1) < 1e-6
    # passthrough off: nothing is called 'ppo'; the rule's default then is the first available algorithm, like get_action's
    # fallback (:757-759), so that update(batch) trains what acts (tests/test_schedule_gpu.py checks the parameters move)
    only = MultiAlgorithmAgent(10, 2, {"tvc_native": {"batch_size": 8, "max_act_rows": 16, "passthrough": False}}, seed=3)
    assert list(only.algorithms) == ["sac"] and only.select_algorithm() == "sac"
    assert only.get_action(obs)[1]["algorithm"] == "sac" and set(only.update(batch)) >= {"q1_loss", "q2_loss", "policy_loss"}


@pytest.mark.parametrize("M,N,K", [(256, 512, 256), (300, 200, 37), (8192, 256, 256), (64, 12, 512), (1000, 256, 12)])
def test_fused_linear_kernel_vs_torch_fp32(M, N, K):
    """Both GEMM kernels (64x64 LDS-tiled, skinny split-K) against torch.nn.functional.linear in fp32, ragged
    shapes included; MFMA f32 is an exact fma chain, so only the summation order differs (<= 1e-5 * sqrt(K))."""
    import ctypes as C
    from tvc_ai_amd import _native as nat
    L = nat.load()
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    X = torch.randn(M, K, device="cuda", generator=g)
    W = torch.randn(N, K, device="cuda", generator=g) / K ** 0.5
    b = torch.randn(N, device="cuda", generator=g)
    for act, fn in ((0, lambda t: t), (1, torch.nn.functional.gelu), (2, torch.relu)):
        ref = fn(torch.nn.functional.linear(X, W, b))
        for variant in (1, 3):
            if variant == 3 and M * N > 512 * 512:
                continue
            Y = torch.full((M, N), float("nan"), device="cuda")
            nat.check(L.tvc_nn_linear_forward(X.data_ptr(), W.data_ptr(), b.data_ptr(), Y.data_ptr(), M, N, K, act, variant,
                                              torch.cuda.current_stream().cuda_stream))
            err = (Y - ref).abs().max().item()
            assert err <= 1e-5 * max(1.0, K ** 0.5), (M, N, K, act, variant, err)


def test_checkpoint_layout_and_round_trip(tmp_path):
    """save_checkpoint writes the reference's file layout (agent/multi_algorithm_agent.py:1098-1141): every key, shape and
    dtype of the MANIFEST taken from a file the reference's own save_checkpoint wrote (tests/golden/ckpt_ref_manifest.json),
    optimizer_{policy,q1,q2}_state in torch.optim.Adam.state_dict() layout included; load -> identical deterministic actions
    (the reference's own integration test asks for atol 1e-6), identical moments and step counters.  The reference's own
    load_checkpoint reading such a file is tested in the build container (tests/test_checkpoint_cpu.py)."""
    import collections
    from tvc_ai_amd import checkpoint as ckpt
    from tvc_ai_amd.agent import MultiAlgorithmAgent
    man = json.load(open(os.path.join(HERE, "golden", "ckpt_ref_manifest.json")))
    cfg = {"tvc_native": {"batch_size": 32, "max_act_rows": 64}}
    a = MultiAlgorithmAgent(10, 2, cfg, seed=1)
    g = torch.Generator().manual_seed(0)
    mk = lambda: {"states": torch.randn(32, 10, generator=g), "actions": torch.rand(32, 2, generator=g) * 2 - 1,
                  "rewards": torch.randn(32, generator=g), "next_states": torch.randn(32, 10, generator=g),
                  "dones": torch.rand(32, generator=g) < 0.1}
    for _ in range(3):
        assert "error" not in a.update(mk(), algorithm="sac")
    assert "error" not in a.update(mk(), algorithm="ppo")  # the eager pass-through entries travel in the same file
    a.update_performance("sac", np.float64(12.5))  # numpy scalars end up in the deques (scripts/train.py:606)
    path = str(tmp_path / "ckpt.pth")
    a.save_checkpoint(path)
    ck = ckpt.load_file(path)
    assert list(ck.keys()) == man["top_level_keys"] and list(ck["algorithms"]) == man["algorithms"] == ["ppo", "sac", "td3"]
    assert set(ck["algorithms"]["ppo"]) == {"policy_state", "optimizer_state", "type"}
    assert [[k, list(t.shape)] for k, t in ck["algorithms"]["ppo"]["policy_state"].items()] == \
        [[k, sh] for k, sh, _ in man["nets"]["policy"]["state_dict"]]  # PPO's net is the same TransformerPolicyNetwork
    assert isinstance(ck["performance_history"]["sac"], collections.deque) and ck["performance_history"]["sac"].maxlen == 100
    sac = ck["algorithms"]["sac"]
    assert list(sac.keys()) == man["sac_keys"] and sac["type"] == "sac"
    for net, desc in man["nets"].items():
        got = [[k, list(t.shape), str(t.dtype).replace("torch.", "")] for k, t in sac[f"{net}_state"].items()]
        assert got == desc["state_dict"], net
    for opt, desc in man["optimizers"].items():
        od = sac[f"{opt}_state"]
        assert sorted(od["state"].keys()) == sorted(int(i) for i in desc["state"]), opt
        grp = {k: (list(x) if isinstance(x, (list, tuple)) else x) for k, x in od["param_groups"][0].items()}
        assert grp == desc["param_groups"][0]
        assert all(float(st["step"]) == 3.0 for st in od["state"].values())
        for i, st in od["state"].items():
            assert [list(st["exp_avg"].shape), "float32"] == desc["state"][str(i)]["exp_avg"], (opt, i)
    # the dead Q/K rows carry zero moments (the reference's gradient there is exactly zero at sequence length 1)
    inproj = sac["optimizer_policy_state"]["state"][2]["exp_avg"]
    assert inproj.shape == (768, 256) and (inproj[:512] == 0).all() and inproj[512:].abs().sum() > 0
    b = MultiAlgorithmAgent(10, 2, cfg, seed=99)
    b.load_checkpoint(path)
    obs = torch.randn(16, 10)
    act_a, _ = a.get_action(obs, deterministic=True, algorithm="sac")
    act_b, _ = b.get_action(obs, deterministic=True, algorithm="sac")
    np.testing.assert_allclose(act_a, act_b, atol=1e-6)
    lay = a.sac.layout
    for name, off, rows, cols in lay.table:
        assert torch.equal(lay.view(a.sac.params, name), lay.view(b.sac.params, name)), name
        if not name.startswith("target_"):
            assert torch.equal(lay.view(a.sac.adam_m, name), lay.view(b.sac.adam_m, name)), name
            assert torch.equal(lay.view(a.sac.adam_v, name), lay.view(b.sac.adam_v, name)), name
    assert b.sac.adam_steps() == [3, 3] and list(b.performance_history["sac"]) == [12.5]
    # the next update of both agents is the same update
    batch = mk()
    la, lb = a.update(batch, algorithm="sac"), b.update(batch, algorithm="sac")
    for pa, pb in zip(a.algorithms["ppo"]["policy"].parameters(), b.algorithms["ppo"]["policy"].parameters()):
        assert torch.equal(pa, pb)
    assert "error" not in la and set(la) == set(lb)
    # a reference-written file holds PPO / TD3 entries and a value head with its own weights: they survive a load + save
    sac2 = dict(sac)
    sac2["policy_state"] = collections.OrderedDict(sac["policy_state"])
    sac2["policy_state"]["value_head.0.weight"] = torch.full((512, 256), 0.25)
    torch.save({"algorithms": {"ppo": {"type": "ppo"}, "sac": sac2, "td3": {"type": "td3"}},
                "performance_history": {k: collections.deque([1.0], maxlen=100) for k in ("ppo", "sac", "td3")},
                "algorithm_weights": {"ppo": 1.0, "sac": 1.0, "td3": 1.0}, "config": {}}, path)
    b.load_checkpoint(path)
    b.save_checkpoint(path)
    assert (ckpt.load_file(path)["algorithms"]["sac"]["policy_state"]["value_head.0.weight"] == 0.25).all()


@pytest.mark.parametrize("dropout_p", [0.0, 0.1])
def test_gradients_match_autograd_of_the_restatement(dropout_p):
    """the backward kernels on their own: grads after tvc_sac_critic_grads / tvc_sac_actor_grads vs torch.autograd of the
    oracle's losses (same parameters, batch and noise), before any optimiser step can blur the comparison.
    dropout_p = 0.1: the reference's train-mode update; the oracle applies the kernels' own hash masks (DropMasks), so the
    comparison stays element for element; a second update checks that the masks move on with the update counter."""
    from tvc_ai_amd.agent import NativeSAC, sac_cfg, _ref_to_native_name
    torch.set_num_threads(8)
    rec = _recipe()
    meta = json.load(open(os.path.join(HERE, "golden", "sac_ref_meta.json")))
    rng = np.random.default_rng(meta["seed"])
    nets = ref_nets(rec, meta, rng)
    B = 256
    sac = NativeSAC(sac_cfg(0, batch_size=B, max_act_rows=B, dropout_p=dropout_p), init=False)
    load_into_native(sac, nets)
    s, a, r, s2, d = [torch.from_numpy(x) for x in rec.make_batch(rng)]
    e1 = torch.from_numpy(rng.standard_normal((B, 2)).astype(np.float32))
    e2 = torch.from_numpy(rng.standard_normal((B, 2)).astype(np.float32))
    sg, ag, rg, s2g, dg, e1g, e2g = cuda(s, a, r, s2, d, e1, e2)
    orc = st.SacOracle(nets["policy"], nets["q1"], nets["q2"], batch_pe=False, dropout_p=dropout_p,
                       dropout_seed=int(sac.cfg.dropout_seed))  # the handle's own mask sequence (seeded by its constructor)
    kw = (lambda call, z: dict(call=call, z=z)) if dropout_p > 0 else (lambda call, z: {})

    def close(got, want, what):
        scale = want.abs().max().item()
        err = (got.cpu() - want).abs().max().item()
        assert err <= 2e-4 * scale + 1e-7, (what, err, scale)

    # critic phase
    sac.critic_grads(sg, ag, rg, s2g, dg, e1g)
    with torch.no_grad():
        m2, ls2 = orc.actor_fn(orc.P, s2, B) if dropout_p > 0 else orc.actor_fn(orc.P, s2)
        a2 = m2 + torch.exp(ls2) * e1
        y = r + st.GAMMA * (1 - d) * torch.min(orc.critic_fn(orc.TQ[0], s2, a2, **kw(1, 0)),
                                               orc.critic_fn(orc.TQ[1], s2, a2, **kw(1, 1)))
    q_losses = []
    for i, net in enumerate(("q1", "q2")):
        loss = torch.nn.functional.mse_loss(orc.critic_fn(orc.Q[i], s, a, **kw(2, i)), y)
        q_losses.append(float(loss.detach()))
        keys = list(orc.Q[i].keys())
        grads = torch.autograd.grad(loss, [orc.Q[i][k] for k in keys])
        for k, gr in zip(keys, grads):
            close(sac.grad_view(f"{net}.{k}").reshape(gr.shape), gr, f"{net}.{k}")
        orc.opt_q[i].step(orc.Q[i], dict(zip(keys, grads)))
    np.testing.assert_allclose(sac.losses[:2].cpu().numpy(), q_losses, rtol=2e-4)
    sac.critic_apply()
    # actor phase (through the UPDATED critics, data gradients only)
    sac.actor_grads(sg, e2g)
    mean, ls = orc.actor_fn(orc.P, s)
    std = torch.exp(ls)
    a_new = mean + std * e2
    logp = (-((a_new - mean) ** 2) / (2 * std ** 2) - ls - np.log(np.sqrt(2 * np.pi))).sum(-1)
    ploss = -(torch.min(orc.critic_fn(orc.Q[0], s, a_new, **kw(3, 0)), orc.critic_fn(orc.Q[1], s, a_new, **kw(3, 1)))
              - st.ALPHA * logp).mean()
    names = list(orc.P.keys())
    grads = torch.autograd.grad(ploss, [orc.P[k] for k in names], allow_unused=True)
    dm, checked = sac.cfg.d_model, 0
    for k, gr in zip(names, grads):
        if k.startswith("value_head") or k.startswith("pos_encoding") or gr is None:
            continue
        name, sl = _ref_to_native_name("policy", k)
        if sl == "v_rows":
            assert gr[:2 * dm].abs().max().item() == 0.0  # Q/K projection rows are dead at sequence length 1
            gr = gr[2 * dm:3 * dm]
        close(sac.grad_view(name).reshape(gr.shape), gr, name)
        checked += 1
    assert checked >= 60
    np.testing.assert_allclose(sac.losses[2].item(), float(ploss.detach()), rtol=2e-4)
    if dropout_p > 0:
        # finish the update on both sides, then a whole second update: new masks (the counter moved), same agreement
        sac.actor_apply()
        orc.opt_p.step(orc.P, dict(zip(names, grads)))
        with torch.no_grad():
            for i in range(2):
                for k in orc.TQ[i]:
                    orc.TQ[i][k].copy_(st.TAU * orc.Q[i][k] + (1 - st.TAU) * orc.TQ[i][k])
        orc.updates += 1
        first = sac.losses[:3].cpu().numpy().copy()
        want = orc.update(s, a, r, s2, d, e1, e2)
        got = sac.update(sg, ag, rg, s2g, dg, e1g, e2g)[:3].cpu().numpy()
        np.testing.assert_allclose(got, want, rtol=5e-4)
        assert not np.allclose(got, first, rtol=1e-3)
        # and the masks do what dropout does: about p of the activations are zeroed
        f = st.DropMasks(dropout_p).factor(3, 7, 0, 0, 512, 512)
        assert abs((f == 0).float().mean().item() - dropout_p) < 0.01 and abs(f.mean().item() - 1.0) < 0.01
    sac.close()


def test_acting_in_train_mode_applies_the_reference_dropout_sites():
    """get_action of the reference runs the policy in train mode (no .eval() anywhere, agent/...:765): attention-weight dropout
    (whole heads of V at sequence length 1), dropout1 / FFN dropout / dropout2 of every encoder layer and the two head Dropouts.
    tvc_sac_act flags bit 3 = NativeSAC.act(train_mode=True) against the restatement with the kernels' own hash masks (site base
    300, counter = number of train-mode acting calls so far), element for element; then the statistics of the masks."""
    from tvc_ai_amd.agent import NativeSAC, sac_cfg
    from tests import parity_log
    torch.set_num_threads(8)
    rec = _recipe()
    meta = json.load(open(os.path.join(HERE, "golden", "sac_ref_meta.json")))
    rng = np.random.default_rng(meta["seed"])
    nets = ref_nets(rec, meta, rng)
    n, p = 300, 0.1
    sac = NativeSAC(sac_cfg(0, batch_size=64, max_act_rows=512, dropout_p=p), init=False)
    load_into_native(sac, nets)
    P = {k: v.clone() for k, v in nets["policy"].items()}
    obs = torch.from_numpy(rng.standard_normal((n, 10)).astype(np.float32))
    eps = torch.from_numpy(rng.standard_normal((n, 2)).astype(np.float32))
    og, eg = cuda(obs, eps)
    masks = st.DropMasks(p, seed=int(sac.cfg.dropout_seed))
    assert int(sac.cfg.dropout_seed) != 0
    worst = 0.0
    outs = []
    for call in range(3):
        act, mean, ls = sac.act(og, eg, train_mode=True)
        with torch.no_grad():
            m_ref, ls_ref = st.actor_forward(P, obs, False, drop=masks.hook(call, 300))
        a_ref = torch.clamp(m_ref + torch.exp(ls_ref) * eps, -1, 1)
        for got, want in ((mean, m_ref), (ls, ls_ref), (act, a_ref)):
            err = (got.cpu() - want).abs().max().item()
            worst = max(worst, err)
            assert err <= 2e-4 * max(1.0, want.abs().max().item()), (call, err)
        outs.append(mean.cpu().clone())
    assert (outs[0] - outs[1]).abs().max().item() > 1e-3  # fresh masks every call
    # the eval-mode pass is a different function of the same weights, and the train-mode outputs scatter around it
    _, mean_eval, _ = sac.act(og, eg)
    with torch.no_grad():
        m_eval_ref, _ = st.actor_forward(P, obs, False)
    assert (mean_eval.cpu() - m_eval_ref).abs().max().item() <= 2e-4 * max(1.0, m_eval_ref.abs().max().item())
    assert (outs[0] - mean_eval.cpu()).abs().max().item() > 1e-3
    # a handle without dropout refuses the flag instead of silently acting in eval mode
    sac0 = NativeSAC(sac_cfg(0, batch_size=64, max_act_rows=512, dropout_p=0.0), init=False)
    with pytest.raises(Exception):
        sac0.act(og, eg, train_mode=True)
    sac0.close()
    # the constructor seed changes the mask sequence, and the call counter travels (checkpoints)
    assert sac.act_counter() == 3
    other = NativeSAC(sac_cfg(0, batch_size=64, max_act_rows=512, dropout_p=p), init=False, seed=77)
    load_into_native(other, nets)
    other.set_act_counter(2)
    _, m_other, _ = other.act(og, eg, train_mode=True)  # same weights, same call index as outs[2], another seed
    assert int(other.cfg.dropout_seed) != int(sac.cfg.dropout_seed) and (m_other.cpu() - outs[2]).abs().max().item() > 1e-3
    sac.set_act_counter(2)
    _, m_again, _ = sac.act(og, eg, train_mode=True)    # the same handle at the same call index replays its masks
    assert torch.equal(m_again.cpu(), outs[2])
    other.close()
    sac.close()
    parity_log.record("test_acting_in_train_mode_applies_the_reference_dropout_sites", rows=n, calls=3, worst_abs_err=worst)


@pytest.mark.parametrize("M,N,K,act", [(8192, 256, 512, 0), (6145, 256, 256, 0), (7000, 512, 256, 1), (6400, 512, 512, 1)])
def test_fused_linear_layernorm_kernel_vs_torch_fp32(M, N, K, act):
    """the acting-pass kernel (Linear + act + residual + LayerNorm in one launch, 32 complete rows per workgroup) on its own"""
    from tvc_ai_amd import _native as nat
    L = nat.load()
    g = torch.Generator().manual_seed(M + N + K)
    X, W = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / K ** 0.5
    b, R = 0.1 * torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    gam, bet = 1 + 0.1 * torch.randn(N, generator=g), 0.1 * torch.randn(N, generator=g)
    z = torch.nn.functional.linear(X, W, b)
    z = torch.nn.functional.gelu(z) if act == 1 else z
    want = torch.nn.functional.layer_norm(z + R, (N,), gam, bet, 1e-5)
    Xg, Wg, bg, Rg, gg, beg = cuda(X, W, b, R, gam, bet)
    Y = torch.empty(M, N, device="cuda")
    nat.check(L.tvc_nn_linear_ln_forward(Xg.data_ptr(), Wg.data_ptr(), bg.data_ptr(), Rg.data_ptr(), gg.data_ptr(), beg.data_ptr(),
                                         Y.data_ptr(), M, N, K, act, torch.cuda.current_stream().cuda_stream))
    np.testing.assert_allclose(Y.cpu().numpy(), want.numpy(), atol=3e-5, rtol=1e-5)
