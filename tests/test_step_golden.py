"""CPU oracle vs goldens of the reference's OWN reset()/step() (tests/golden/gen_step_golden.py: the reference module driven
over a scripted, recording pybullet stand-in).  Pins SURVEY section 8 rows a1 (step ordering), a2/a3 (the wrench the reference
assembles), a6/a7 (derived scalars, observation and its one-step phase lag), a11 (curiosity timing), a12-a14 and the constants of
reset(); p.stepSimulation itself (a4/a5) stays unpinned.  Runs without a GPU and without /root/reference."""
import importlib.util
import json
import os

import numpy as np
import pytest
import torch

from oracle import envoracle as eo
from oracle import sac_torch as st

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")
SCENARIOS = ["nominal", "hover", "success", "burnout"]


def _mod(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(GOLD, name + ".py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def curiosity_weights():
    rec, gs = _mod("gen_sac_golden"), _mod("gen_step_golden")
    vals = rec.fill_params(gs.curiosity_named(), np.random.default_rng(gs.CUR_SEED))
    return {k: torch.from_numpy(v) for k, v in vals.items()}


def load(name):
    return np.load(os.path.join(GOLD, f"step_ref_{name}.npz"))


def meta():
    with open(os.path.join(GOLD, "step_ref_meta.json")) as f:
        return json.load(f)


def test_reset_constants_match_what_the_reference_passes_to_pybullet():
    """_setup_physics / _create_enhanced_rocket (env/...:324-352, 409-464) as recorded by the stand-in."""
    c = meta()["constants"]
    p = eo.default_params()
    assert p.mass == c["body"]["mass"] == 2.0
    assert list(p.init_pos) == c["body"]["position"] and list(p.init_quat) == c["body"]["orientation"]
    assert list(p.inertia) == c["body_dynamics"]["localInertiaDiagonal"]  # bit-equal: python evaluation order kept
    assert p.lin_damp == c["body_dynamics"]["linearDamping"] and p.ang_damp == c["body_dynamics"]["angularDamping"]
    assert p.gravity == -c["gravity"][2] and c["gravity"][:2] == [0, 0]
    assert p.n_sub == c["engine"]["numSubSteps"] and p.dt_sub * p.n_sub == c["engine"]["fixedTimeStep"]
    assert p.radius == c["collision"]["radius"] and 2 * p.half_len == c["collision"]["height"]
    assert p.mu == c["plane_dynamics"]["lateralFriction"] * c["body_dynamics"]["lateralFriction"]
    assert p.thrust == meta()["thrust_profile"]
    assert meta()["phases"] == eo.PHASE_NAMES


@pytest.mark.parametrize("name", SCENARIOS)
def test_oracle_step_reproduces_the_reference_step(name):
    g = load(name)
    T = g["action"].shape[0]
    Fm = curiosity_weights()
    env = eo.OracleEnv(contact=1, auto_reset=0, distinct_window=1000, max_episode_steps=int(g["max_episode_steps"]))
    prev_obs8 = None
    worst = dict(F=0.0, tau=0.0, obs=0.0, reward=0.0, comps=0.0, curiosity=0.0, info=0.0, post=0.0)
    for t in range(T):
        if g["reset_before"][t]:
            obs0 = env.reset()
            assert np.array_equal(obs0, g["reset_obs"][t]), t  # reset() observation (:404), bit-equal float32
            prev_obs8 = None
        a_in = g["action"][t]
        a = np.clip(a_in, -1.0, 1.0)
        env.p.thrust = float(g["thrust_t"][t])
        pre = env.state13()
        # ---- a2/a3: the wrench the reference handed to pybullet, summed about the COM
        F, tau = env.wrench(a_in)
        F_ref = g["f_grav"][t].copy()
        tau_ref = g["t_aero"][t].copy()
        has_thrust = not np.isnan(g["f_thrust"][t, 0])
        assert has_thrust == (env.e.fuel > 0.0), t  # thrust test uses the fuel BEFORE the decrement (:530)
        if has_thrust:
            F_ref += g["f_thrust"][t]
            tau_ref += np.cross(g["p_thrust"][t] - pre[0:3], g["f_thrust"][t])  # applyExternalForce at a world point
        if not np.isnan(g["f_drag"][t, 0]):
            F_ref += g["f_drag"][t]
        assert np.isnan(g["f_drag"][t, 0]) == (not np.linalg.norm(pre[7:10]) > 0.1), t  # drag switch (:572)
        worst["F"] = max(worst["F"], float(np.abs(F - F_ref).max()))
        worst["tau"] = max(worst["tau"], float(np.abs(tau - tau_ref).max()))
        # ---- a1: the full step
        out = env.step(a_in)
        worst["post"] = max(worst["post"], float(np.abs(env.state13() - g["post"][t]).max()))
        obs = np.frombuffer(out.obs, dtype=np.float32).copy()
        worst["obs"] = max(worst["obs"], float(np.abs(obs - g["obs"][t]).max()))
        assert obs[8] == g["obs"][t, 8], (t, "observation phase lags the phase update by one step (:482 vs :485)")
        comps = g["comps"][t]
        cur = 0.0 if np.isnan(comps[9]) else comps[9]
        assert np.isnan(comps[9]) == (prev_obs8 is None), (t, "curiosity is skipped on the first step of an episode (:496)")
        if prev_obs8 is not None:
            with torch.no_grad():
                c = float(st.curiosity_reward(Fm, torch.from_numpy(prev_obs8[None]), torch.from_numpy(a[None].astype(np.float32)),
                                              torch.from_numpy(obs[None, :8])))
            worst["curiosity"] = max(worst["curiosity"], abs(c - cur) / max(abs(cur), 1e-12))
        prev_obs8 = obs[:8].copy()
        # extrinsic reward: the reference adds the bonus AFTER the clip and does not store it in the reward history
        worst["reward"] = max(worst["reward"], abs(out.reward - (g["reward"][t] - cur)) / max(1.0, abs(out.reward)))
        for k in range(9):
            ref_c = 0.0 if np.isnan(comps[k]) else comps[k]
            worst["comps"] = max(worst["comps"], abs(out.components[k] - ref_c) / max(1.0, abs(ref_c)))
        assert bool(out.terminated) == bool(g["term"][t]) and bool(out.truncated) == bool(g["trunc"][t]), t
        # ---- a13: info
        assert env.e.phase == g["info_phase"][t] and bool(env.e.mission_successful) == bool(g["info_success"][t]), t
        assert env.e.step == g["info_step"][t], t
        assert (env.e.success_run >= 10) == bool(g["info_criteria10"][t]), t
        sc = out.sc
        for mine, ref in ((sc.altitude, g["info_altitude"][t]), (np.degrees(sc.tilt), g["info_tilt_deg"][t]),
                          (sc.omega_mag, g["info_omega"][t]), (env.e.fuel, g["info_fuel"][t])):
            worst["info"] = max(worst["info"], abs(mine - ref) / max(1.0, abs(ref)))
        assert np.array_equal(g["info_position"][t], g["post"][t, 0:3])
    print(name, worst)
    assert worst["post"] < 1e-12           # the script IS the oracle's trajectory (same C code, same flags)
    assert worst["F"] < 1e-12 and worst["tau"] < 1e-12
    assert worst["obs"] <= 1.2e-7          # float32(double): scipy-vs-own helper differences of 1 ulp(double) can flip a rounding
    assert worst["reward"] < 1e-12 and worst["comps"] < 1e-12 and worst["info"] < 1e-12
    assert worst["curiosity"] < 5e-5       # fp32 torch forward model vs the reference's own (same weights)


def test_step_golden_covers_the_branches_it_claims():
    g = {n: load(n) for n in SCENARIOS}
    assert g["nominal"]["term"].sum() >= 5 and (g["nominal"]["info_altitude"] < 0.55).any()        # ground contact + terminations
    assert (np.abs(g["nominal"]["action"]) > 1.0).any()                                             # np.clip(action) exercised
    assert g["nominal"]["reset_before"][150] == 1 and not g["nominal"]["term"][149]                 # reset in mid-flight
    assert (np.nan_to_num(g["nominal"]["comps"][:, 6]) == -1000.0).any()                            # crash penalty
    assert g["hover"]["trunc"].sum() >= 2 and set(g["hover"]["info_phase"].tolist()) >= {0, 1, 2, 3}
    assert g["hover"]["info_phase"][198] == 0 and g["hover"]["info_phase"][199] == 1                # fuel < 0.8 fires at step 200
    assert g["success"]["info_success"].sum() > 100 and g["success"]["term"][99] == 1 and g["success"]["term"][100] == 1
    assert np.isnan(g["burnout"]["f_thrust"][:, 0]).sum() > 20 and np.isnan(g["burnout"]["f_thrust"][1000, 0])
    assert not np.isnan(g["burnout"]["f_thrust"][999, 0])                                           # last thrusted step = 1000th


def test_quaternion_helpers_match_scipy():
    """oracle quat->matrix / quat->euler (restated pybullet helpers, ASSUMPTION(bullet)) vs an independent implementation."""
    from scipy.spatial.transform import Rotation
    import ctypes as C
    L = eo.lib()
    rng = np.random.default_rng(5)
    q = rng.standard_normal((2000, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    m, e = (C.c_double * 9)(), (C.c_double * 3)()
    worst_m = worst_e = 0.0
    for i in range(q.shape[0]):
        qi = np.ascontiguousarray(q[i])
        L.tvc_oracle_quat_to_matrix(qi.ctypes.data_as(C.POINTER(C.c_double)), m)
        L.tvc_oracle_quat_to_euler(qi.ctypes.data_as(C.POINTER(C.c_double)), e)
        R = Rotation.from_quat(qi)
        worst_m = max(worst_m, float(np.abs(np.array(m).reshape(3, 3) - R.as_matrix()).max()))
        if abs(-2.0 * (qi[0] * qi[2] - qi[3] * qi[1])) < 0.999:  # away from the gimbal-lock branch
            d = np.array(e) - R.as_euler("xyz")
            d = (d + np.pi) % (2 * np.pi) - np.pi
            worst_e = max(worst_e, float(np.abs(d).max()))
    assert worst_m < 1e-14 and worst_e < 1e-11, (worst_m, worst_e)
