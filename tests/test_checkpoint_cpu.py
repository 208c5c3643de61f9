"""Reference checkpoint layout (agent/multi_algorithm_agent.py:1098-1179) without a GPU: tvc_ai_amd.checkpoint packs the
learner's flat buffers into the reference's 'sac' entry and back.

  * against the MANIFEST of a file written by the reference's own save_checkpoint (tests/golden/ckpt_ref_manifest.json,
    generator tests/golden/gen_ckpt_manifest.py): key order, shapes, dtypes, optimizer param_groups and which parameters
    hold optimizer state;
  * exact round trip pack -> unpack;
  * in the build container only (needs /root/reference): the REFERENCE'S OWN load_checkpoint reads a file packed here
    (strict load_state_dict + Adam.load_state_dict), and a file the reference's save_checkpoint writes loads here through the
    weights-only loader (performance_history is a dict of deques).
"""
import collections
import json
import os
import sys

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def layout_and_buffers(seed=0):
    from tvc_ai_amd import agent as ag
    from tvc_ai_amd import checkpoint as ck
    cfg = ag.sac_cfg(0)
    table = ag.tensor_table(cfg)  # host-only query of the C library
    lay = ck.Layout(table, 0, cfg.d_model, cfg.n_layers)
    L = ag.nat.load()
    import ctypes as C
    n_all, n_tr = L.tvc_sac_param_count(C.byref(cfg)), L.tvc_sac_trainable_count(C.byref(cfg))
    g = torch.Generator().manual_seed(seed)
    params = torch.randn(n_all, generator=g) * 0.05
    m = torch.randn(n_tr, generator=g) * 1e-3
    v = torch.rand(n_tr, generator=g) * 1e-4
    passive = ck.default_passive(cfg.d_model, cfg.n_layers, cfg.head1, cfg.head2, seed)
    return lay, params, m, v, passive, cfg


def same_tensors(lay, a, b, trainable_only=False):
    """flat buffers agree on every tensor of the table (alignment padding between tensors is not state)"""
    for name, off, rows, cols in lay.table:
        if trainable_only and (name.startswith("target_") or off + rows * cols > a.numel()):
            continue
        if not torch.equal(lay.view(a, name), lay.view(b, name)):
            return False
    return True


def manifest():
    with open(os.path.join(HERE, "golden", "ckpt_ref_manifest.json")) as f:
        return json.load(f)


def test_packed_entry_matches_the_reference_manifest():
    from tvc_ai_amd import checkpoint as ck
    lay, params, m, v, passive, _ = layout_and_buffers()
    entry = ck.pack_sac(lay, params, m, v, [5, 5], passive)
    man = manifest()
    assert list(entry.keys()) == man["sac_keys"] and entry["type"] == man["sac_type"] == "sac"
    for net, desc in man["nets"].items():
        sd = entry[f"{net}_state"]
        assert [[k, list(t.shape), str(t.dtype).replace("torch.", "")] for k, t in sd.items()] == desc["state_dict"], net
    for opt, desc in man["optimizers"].items():
        od = entry[f"{opt}_state"]
        assert list(od.keys()) == ["state", "param_groups"]
        grp = {k: (list(x) if isinstance(x, (list, tuple)) else x) for k, x in od["param_groups"][0].items()}
        assert grp == desc["param_groups"][0], opt
        assert sorted(od["state"].keys()) == sorted(int(i) for i in desc["state"]), opt   # value_head: no state, like the reference
        for i, st in od["state"].items():
            ref = desc["state"][str(i)]
            assert list(st.keys()) == list(ref.keys())
            for k in ("exp_avg", "exp_avg_sq", "step"):
                assert [list(st[k].shape), str(st[k].dtype).replace("torch.", "")] == ref[k], (opt, i, k)
    names = man["nets"]["policy"]["parameters"]
    missing = [names[i] for i in range(len(names)) if i not in entry["optimizer_policy_state"]["state"]]
    assert missing == man["optimizers"]["optimizer_policy"]["params_without_state"]


def test_pack_unpack_round_trip_is_exact():
    from tvc_ai_amd import checkpoint as ck
    lay, params, m, v, passive, _ = layout_and_buffers(3)
    entry = ck.pack_sac(lay, params, m, v, [7, 6], passive)
    p2, m2, v2 = torch.zeros_like(params), torch.ones_like(m), torch.ones_like(v)
    steps, passive2, have = ck.unpack_sac(lay, entry, p2, m2, v2)
    assert have and steps == [7, 6]
    assert same_tensors(lay, p2, params) and same_tensors(lay, m2, m, True) and same_tensors(lay, v2, v, True)
    for k in ("policy.value_head.0.weight", "policy.transformer_encoder.layers.2.self_attn.in_proj_weight"):
        assert torch.equal(passive2[k][:512 if "in_proj" in k else None], passive[k][:512 if "in_proj" in k else None])
    # no Adam step yet -> empty optimizer state, like a fresh torch.optim.Adam
    e0 = ck.pack_sac(lay, params, m, v, [0, 0], passive)
    assert e0["optimizer_q1_state"]["state"] == {} and e0["optimizer_policy_state"]["state"] == {}


def test_weights_only_loader_accepts_the_references_deques(tmp_path):
    from tvc_ai_amd import checkpoint as ck
    path = str(tmp_path / "c.pth")
    import numpy as np
    torch.save({"performance_history": {"sac": collections.deque([1.0, np.float64(2.0)], maxlen=100)},
                "w": collections.OrderedDict(a=torch.ones(2))}, path)
    with pytest.raises(Exception):
        torch.load(path, weights_only=True)  # what round 1 did: refuses collections.deque
    c = ck.load_file(path)
    assert list(c["performance_history"]["sac"]) == [1.0, 2.0] and torch.equal(c["w"]["a"], torch.ones(2))
    # the fallback loader resolves data constructors only: a file that names anything else is refused, not executed
    import pickle

    class Evil:
        def __reduce__(self):
            return (os.system, ("echo pwned > %s" % (tmp_path / "pwned"),))
    torch.save({"performance_history": {"sac": collections.deque([1.0])}, "x": Evil()}, path)
    with pytest.raises(pickle.UnpicklingError, match="only tensors"):
        ck.load_file(path)
    assert not (tmp_path / "pwned").exists()


@pytest.mark.skipif(not os.path.isdir(REF), reason="needs the reference tree (build container only)")
def test_reference_load_checkpoint_reads_a_file_packed_here_and_back(tmp_path, monkeypatch):
    import functools
    import yaml
    from tvc_ai_amd import checkpoint as ck
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    try:
        from agent.multi_algorithm_agent import MultiAlgorithmAgent as RefAgent  # the reference, unmodified
    finally:
        sys.path.remove(REF)
    lay, params, m, v, passive, _ = layout_and_buffers(11)
    entry = ck.pack_sac(lay, params, m, v, [9, 8], passive)
    path = str(tmp_path / "build.pth")
    torch.save({"algorithms": {"sac": entry}, "performance_history": {k: collections.deque([3.0], maxlen=100) for k in ("ppo", "sac", "td3")},
                "algorithm_weights": {"sac": 1.0}, "config": {"note": "written by tvc_ai_amd"}}, path)
    cfg = yaml.safe_load(open(os.path.join(REF, "config", "config.yaml")))
    cfg["hardware"] = {"device": "cpu"}
    ref = RefAgent(10, 2, cfg)
    ref.device = torch.device("cpu")
    # the reference calls torch.load(path, map_location=...) (:1145), written for torch < 2.6 where that meant
    # weights_only=False; on this image's torch 2.10 it could not even re-read its OWN files (deque in performance_history).
    # Give it the default it was written for -- the file is one this test wrote itself a few lines above.
    monkeypatch.setattr(torch, "load", functools.partial(torch.load, weights_only=False))
    ref.load_checkpoint(path)  # strict nn.Module.load_state_dict x5 + Adam.load_state_dict x3: raises on any mismatch
    monkeypatch.undo()
    sac = ref.algorithms["sac"]
    for net in ck.NETS:
        for k, t in sac[net].state_dict().items():
            assert torch.equal(t, entry[f"{net}_state"][k]), (net, k)
    st = sac["optimizer_q1"].state_dict()["state"]
    assert float(st[0]["step"]) == 9.0 and torch.equal(st[4]["exp_avg"], entry["optimizer_q1_state"]["state"][4]["exp_avg"])
    assert sac["optimizer_policy"].state_dict()["param_groups"][0]["lr"] == 3e-4
    # ... and the other direction: the reference writes, the weights-only loader + unpack read
    back = str(tmp_path / "ref.pth")
    ref.save_checkpoint(back)
    c = ck.load_file(back)
    assert isinstance(c["performance_history"]["sac"], collections.deque) and set(c["algorithms"]) == {"ppo", "sac", "td3"}
    p2, m2, v2 = torch.zeros_like(params), torch.zeros_like(m), torch.zeros_like(v)
    steps, _, have = ck.unpack_sac(lay, c["algorithms"]["sac"], p2, m2, v2)
    assert have and steps == [9, 8]
    assert same_tensors(lay, p2, params) and same_tensors(lay, m2, m, True) and same_tensors(lay, v2, v, True)


@pytest.mark.skipif(not os.path.isdir(REF), reason="needs the reference tree (build container only)")
def test_passthrough_nets_are_the_references_nets():
    """The eager PPO / TD3 pass-through (tvc_ai_amd/passthrough.py) against the reference's own modules: identical state_dict
    keys and shapes (strict load both ways) and identical forward outputs in eval mode on the same weights, batch-row
    positional encoding included."""
    import yaml
    from tvc_ai_amd import passthrough as pt
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    try:
        from agent.multi_algorithm_agent import MultiAlgorithmAgent as RefAgent
    finally:
        sys.path.remove(REF)
    cfg = yaml.safe_load(open(os.path.join(REF, "config", "config.yaml")))
    cfg["hardware"] = {"device": "cpu"}
    ref = RefAgent(10, 2, cfg)
    mine = pt.make_ppo(10, 2, cfg, torch.device("cpu"))
    ref_pol = ref.algorithms["ppo"]["policy"]
    ref_pol.load_state_dict(mine["policy"].state_dict())            # strict
    mine["policy"].load_state_dict(ref_pol.state_dict())
    ref_pol.eval(); mine["policy"].eval()
    x = torch.randn(7, 10)
    with torch.no_grad():
        for a, b in zip(ref_pol(x), mine["policy"](x)):
            assert torch.allclose(a, b, atol=1e-6), (a - b).abs().max()
    assert mine["optimizer"].param_groups[0]["lr"] == ref.algorithms["ppo"]["optimizer"].param_groups[0]["lr"] == 2.5e-4
    td = pt.make_td3(10, 2, torch.device("cpu"))
    for k in ("policy", "q1", "q2", "target_policy", "target_q1", "target_q2"):
        ref.algorithms["td3"][k].load_state_dict(td[k].state_dict())  # strict
    ref.algorithms["td3"]["policy"].eval(); td["policy"].eval()
    with torch.no_grad():
        assert torch.allclose(ref.algorithms["td3"]["policy"](x), td["policy"](x), atol=1e-6)
    # one PPO update on the same batch from the same weights gives the same losses (dropout off on both sides)
    batch = {"states": x, "actions": torch.rand(7, 2) * 2 - 1, "rewards": torch.randn(7), "next_states": x + 0.1, "dones": torch.zeros(7)}
    lr_ = ref._update_ppo(batch)
    lm = pt.update_ppo(mine, batch)
    for k in lr_:
        assert abs(lr_[k] - lm[k]) <= 1e-5 * max(1.0, abs(lr_[k])), (k, lr_[k], lm[k])
