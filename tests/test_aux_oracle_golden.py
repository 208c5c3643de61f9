"""Torch restatements of the curiosity bonus and the safety layer (oracle/sac_torch.py) vs goldens produced by the
reference's own classes (tests/golden/gen_aux_golden.py)."""
import importlib.util
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


def aux_setup():
    rec, aux = _mod("gen_sac_golden"), _mod("gen_aux_golden")
    g = np.load(os.path.join(HERE, "golden", "aux_ref.npz"))
    rng = np.random.default_rng(aux.SEED)
    cur_named = [("0.weight", (256, 10)), ("0.bias", (256,)), ("2.weight", (256, 256)), ("2.bias", (256,)),
                 ("4.weight", (8, 256)), ("4.bias", (8,))]
    Fm = {k: torch.from_numpy(v) for k, v in rec.fill_params(cur_named, rng).items()}
    cs, ca, cs2 = aux.make_inputs(rng)
    saf_named = [("0.weight", (128, 12)), ("0.bias", (128,)), ("2.weight", (64, 128)), ("2.bias", (64,)),
                 ("4.weight", (2, 64)), ("4.bias", (2,))]
    Sm = {k: torch.from_numpy(v) for k, v in rec.fill_params(saf_named, rng).items()}
    ss, sa, ss2 = aux.make_inputs(rng)
    return g, Fm, (cs, ca, cs2), Sm, (ss, sa, ss2)


def test_curiosity_and_safety_match_reference():
    g, Fm, (cs, ca, cs2), Sm, (ss, sa, _) = aux_setup()
    with torch.no_grad():
        r = st.curiosity_reward(Fm, torch.from_numpy(cs[:64, :8]), torch.from_numpy(np.clip(ca[:64], -1, 1)),
                                torch.from_numpy(cs2[:64, :8]))
        out, viol = st.safety_layer(Sm, torch.from_numpy(ss), torch.from_numpy(sa))
    np.testing.assert_allclose(r.numpy(), g["cur_reward"], rtol=2e-5, atol=1e-8)
    np.testing.assert_allclose(out.numpy(), g["safety_out"], atol=2e-6)
    assert 0.5 < viol.float().mean() < 1.0
