"""Oracle (oracle/tvc_oracle.c) vs golden vectors taken from the reference's own Python
(tests/golden/gen_env_golden.py): reward terms, mission phase, success window, termination.
ref: env/enhanced_rocket_tvc_env.py:86-224, 635-721."""
import glob
import os

import numpy as np
import pytest

from oracle import envoracle as eo

SCENARIOS = ["random", "benign", "success_window", "touchdown", "antihack", "terminations"]


@pytest.mark.parametrize("name", SCENARIOS)
def test_logic_matches_reference(golden_dir, name):
    g = np.load(os.path.join(golden_dir, f"env_logic_{name}.npz"))
    env = eo.OracleEnv(max_episode_steps=int(g["max_episode_steps"]), distinct_window=1000)
    T = len(g["reward"])
    worst = 0.0
    for t in range(T):
        if g["reset_before"][t]:
            env.reset()
        # fuel / step bookkeeping that the physics half does (ref :530-533, :478)
        if env.e.fuel > 0:
            env.e.fuel = max(0.0, env.e.fuel - 0.001)
        env.e.step += 1
        assert env.e.fuel == g["fuel"][t]
        assert env.e.step == g["step"][t]
        s = g["scalars"][t]
        sc = eo.Scalars(altitude=s[0], tilt=s[1], omega_mag=s[2], v_h=s[3], v_z_abs=s[4], x=s[5], y=s[6],
                        crashed=int(s[0] < 0.1))
        out = env.logic(sc, g["actions"][t])
        assert env.e.phase == g["phase"][t], (name, t)
        assert env.e.mission_successful == g["success"][t], (name, t)
        assert out.terminated == g["term"][t], (name, t)
        assert out.truncated == g["trunc"][t], (name, t)
        comps = np.array(out.components[:9])
        np.testing.assert_allclose(comps, g["comps"][t], rtol=1e-14, atol=1e-14, err_msg=f"{name} t={t}")
        ref = g["reward"][t]
        err = abs(out.reward - ref) / max(1.0, abs(ref))
        worst = max(worst, err)
        assert err <= 1e-13, (name, t, out.reward, ref)
    print(name, "worst rel err", worst)


def test_fuel_sequence_crosses_at_200():
    # ref :642 BOOST->COAST when fuel < 0.8; sequential fp64 subtraction crosses at step 200
    L = eo.lib()
    assert L.tvc_oracle_fuel_after(199) >= 0.8
    assert L.tvc_oracle_fuel_after(200) < 0.8
    assert L.tvc_oracle_fuel_after(200) == 0.7999999999999998
