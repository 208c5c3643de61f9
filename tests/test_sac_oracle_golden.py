"""The eager-PyTorch SAC restatement (oracle/sac_torch.py) vs golden vectors produced by the reference's own
agent code (tests/golden/gen_sac_golden.py): actor/critic forward, physics loss, three SAC updates.
fp32 torch on both sides; tolerances: forward 2e-5 abs, losses 2e-5 rel, parameter digests 1e-4 rel."""
import importlib.util
import json
import os

import numpy as np
import pytest
import torch

from oracle import sac_torch as st

HERE = os.path.dirname(os.path.abspath(__file__))


def _recipe():
    spec = importlib.util.spec_from_file_location("gen_sac_golden", os.path.join(HERE, "golden", "gen_sac_golden.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.fixture(scope="module")
def setup():
    torch.set_num_threads(4)
    rec = _recipe()
    meta = json.load(open(os.path.join(HERE, "golden", "sac_ref_meta.json")))
    g = np.load(os.path.join(HERE, "golden", "sac_ref.npz"))
    rng = np.random.default_rng(meta["seed"])
    nets = {}
    for k in ("policy", "q1", "q2"):
        named = [(n, tuple(s)) for n, s in meta["nets"][k]]
        nets[k] = {n: torch.from_numpy(v) for n, v in rec.fill_params(named, rng).items()}
    return rec, meta, g, rng, nets


def test_forward_and_updates_match_reference(setup):
    rec, meta, g, rng, nets = setup
    s, a, r, s2, d = [torch.from_numpy(x) for x in rec.make_batch(rng)]
    with torch.no_grad():
        m_b, ls_b = st.actor_forward(nets["policy"], s, batch_pe=True)
        m_0, ls_0 = st.actor_forward(nets["policy"], s[:8], batch_pe=False)
        q1 = st.critic_forward(nets["q1"], s, a)
        q2 = st.critic_forward(nets["q2"], s, a)
    np.testing.assert_allclose(m_b.numpy(), g["fwd_mean_batchpe"], atol=2e-5, rtol=0)
    np.testing.assert_allclose(ls_b.numpy(), g["fwd_logstd_batchpe"], atol=2e-5, rtol=0)
    np.testing.assert_allclose(m_0.numpy(), g["fwd_mean_pe0_first8"], atol=2e-5, rtol=0)   # PE(0) == reference at B=1
    np.testing.assert_allclose(ls_0.numpy(), g["fwd_logstd_pe0_first8"], atol=2e-5, rtol=0)
    np.testing.assert_allclose(q1.numpy(), g["fwd_q1"], atol=2e-5, rtol=0)
    np.testing.assert_allclose(q2.numpy(), g["fwd_q2"], atol=2e-5, rtol=0)
    pl, parts = st.physics_loss(s, a, s2)
    np.testing.assert_allclose([float(pl)] + [float(p) for p in parts], g["physics_loss"], rtol=1e-5)
    with torch.no_grad():
        det = torch.clamp(st.actor_forward(nets["policy"], s[:16], batch_pe=True)[0], -1, 1)
    np.testing.assert_allclose(det.numpy(), g["get_action_det"], atol=2e-5)

    sac = st.SacOracle(nets["policy"], nets["q1"], nets["q2"], batch_pe=True)
    for u in range(3):
        e1 = torch.from_numpy(rng.standard_normal((256, 2)).astype(np.float32))
        e2 = torch.from_numpy(rng.standard_normal((256, 2)).astype(np.float32))
        if u > 0:
            s, a, r, s2, d = [torch.from_numpy(x) for x in rec.make_batch(rng)]
        l1, l2, lp = sac.update(s, a, r, s2, d, e1, e2)
        np.testing.assert_allclose([l1, l2, lp], g["losses"][u, :3], rtol=2e-5, err_msg=f"update {u}")
        if u in (0, 2):
            groups = {"policy": sac.P, "q1": sac.Q[0], "q2": sac.Q[1], "target_q1": sac.TQ[0], "target_q2": sac.TQ[1]}
            for k, params in groups.items():
                dig = np.stack([rec.tensor_digest(p) for p in params.values()])
                ref = g[f"u{u}_{k}_digest"]
                scale = np.maximum(np.abs(ref[:, 1:2]), 1e-3)  # |.|-sum of the tensor sets the scale of sum / sumsq
                assert np.all(np.abs(dig[:, :3] - ref[:, :3]) <= 1e-4 * np.maximum(scale, np.abs(ref[:, :3]))), (u, k)
                # single elements: Adam's first steps are sign-like (lr*g/(|g|+eps)), so an element whose gradient
                # is ~0 can legitimately land anywhere within lr per step
                np.testing.assert_allclose(dig[:, 3:], ref[:, 3:], rtol=2e-4, atol=st.LR * (u + 1) * 1.01,
                                           err_msg=f"{u} {k}")
            m_dig = np.stack([rec.tensor_digest(sac.opt_p.m[k]) for k in sac.P])
            ref = g[f"u{u}_optimizer_policy_expavg_digest"]
            live = np.abs(ref[:, 1]) > 0
            np.testing.assert_allclose(m_dig[live, 1], ref[live, 1], rtol=2e-3)


def test_value_head_and_qk_rows_are_dead_for_sac(setup):
    """at seq-len 1 the Q/K projections and the value head never influence a SAC update."""
    rec, meta, g, rng, nets = setup
    ref = g["u2_policy_digest"]
    names = [n for n, _ in meta["nets"]["policy"]]
    rng2 = np.random.default_rng(meta["seed"])
    init = rec.fill_params([(n, tuple(s)) for n, s in meta["nets"]["policy"]], rng2)
    for i, n in enumerate(names):
        if n.startswith("value_head"):
            np.testing.assert_allclose(ref[i], rec.tensor_digest(torch.from_numpy(init[n])), rtol=1e-12)
