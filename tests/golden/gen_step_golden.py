#!/usr/bin/env python3
"""Golden vectors of the reference's OWN reset()/step() (env/enhanced_rocket_tvc_env.py:381-407, 466-518), driven over a
scripted, recording stand-in for pybullet.

Runs ONLY in the build container (needs /root/reference).  It imports env/enhanced_rocket_tvc_env.py unmodified after
placing build-authored stand-in modules for `pybullet`, `pybullet_data` and `gymnasium` in sys.modules (none is installed
in this image).  The pybullet stand-in (``ScriptedBullet``) does NO physics:

  * getBasePositionAndOrientation / getBaseVelocity / getDynamicsInfo answer from the body's current scripted pose;
  * stepSimulation() swaps in the NEXT pose of the script;
  * applyExternalForce / applyExternalTorque RECORD their arguments;
  * getMatrixFromQuaternion / getEulerFromQuaternion are scipy.spatial.transform.Rotation (an implementation independent
    of oracle/tvc_oracle.c's helpers: row-major matrix of the (x, y, z, w) quaternion; roll-pitch-yaw = extrinsic 'xyz');
  * createMultiBody / changeDynamics / setGravity / setPhysicsEngineParameter record the constants the reference passes.

The script of poses is produced, in lockstep, by the fp64 oracle (oracle/tvc_oracle.c) stepping the same actions, so the
trajectory is one the oracle's full step reproduces; everything the reference's Python does with those poses -- the wrench it
assembles (:520-585), the observation and its one-step phase lag (:482 vs :485), fuel, phase, success window, reward and its
components, the curiosity bonus and its first-step skip (:496-502), termination, info (:723-742, :513-516), reset (:381-407) --
is the REFERENCE'S output.  tests/test_step_golden.py checks tvc_oracle_wrench / tvc_oracle_step against it on the CPU and
tests/test_step_golden_gpu.py checks the HIP env (N = 1 drop-in wrapper and the vector env) against it.

What this does NOT pin: p.stepSimulation itself (rows a4/a5 of SURVEY section 8 stay "parity unpinned").

Output: tests/golden/step_ref_<scenario>.npz + step_ref_meta.json (data only).
"""
import importlib.util
import json
import os
import sys
import types

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CUR_SEED = 4242  # curiosity forward-model weights: gen_sac_golden.fill_params recipe from this numpy seed

COMP_KEYS = ["mission_completion", "safety_compliance", "fuel_efficiency", "stability_bonus", "control_smoothness",
             "altitude_maintenance", "crash_penalty", "excessive_tilt", "control_saturation", "curiosity"]
PHASES = ["boost", "coast", "landing", "touchdown", "hover", "complete", "failed"]


class ScriptedBullet(types.ModuleType):
    """Recording, scripted stand-in for the pybullet module (see the file docstring)."""
    GUI, DIRECT = 1, 2
    GEOM_CYLINDER = 4
    LINK_FRAME, WORLD_FRAME = 1, 2

    def __init__(self):
        super().__init__("pybullet")
        self.constants = {}      # what the reference passed to the world / body set-up calls
        self.calls = []          # wrench calls of the current control step
        self.state = None        # (pos3, quat4, vel3, omega3)
        self.next_state = None
        self.mass = None
        self.n_connect = self.n_step = 0

    # -- world
    def connect(self, mode):
        self.n_connect += 1
        return 0

    def disconnect(self, client=None):
        return None

    def setAdditionalSearchPath(self, path):
        return None

    def setGravity(self, x, y, z):
        self.constants["gravity"] = [x, y, z]

    def setPhysicsEngineParameter(self, **kw):
        self.constants["engine"] = {k: (float(v) if isinstance(v, float) else int(v)) for k, v in kw.items()}

    def loadURDF(self, name):
        self.constants["plane"] = name
        return 0

    def resetSimulation(self, client=None):
        self.state = None

    def createCollisionShape(self, kind, **kw):
        self.constants["collision"] = {"kind": int(kind), **{k: float(v) for k, v in kw.items()}}
        return 0

    def createVisualShape(self, kind, **kw):
        return 0

    def createMultiBody(self, baseMass, baseCollisionShapeIndex, baseVisualShapeIndex, basePosition, baseOrientation, **kw):
        self.mass = float(baseMass)
        self.constants["body"] = {"mass": float(baseMass), "position": [float(x) for x in basePosition],
                                  "orientation": [float(x) for x in baseOrientation]}
        self.state = (tuple(float(x) for x in basePosition), tuple(float(x) for x in baseOrientation), (0.0, 0.0, 0.0),
                      (0.0, 0.0, 0.0))
        return 1

    def changeDynamics(self, body, link, **kw):
        key = "plane_dynamics" if body == 0 else "body_dynamics"
        d = self.constants.setdefault(key, {})
        for k, v in kw.items():
            d[k] = [float(x) for x in v] if isinstance(v, (list, tuple)) else float(v)

    # -- queries
    def getBasePositionAndOrientation(self, body):
        return self.state[0], self.state[1]

    def getBaseVelocity(self, body):
        return self.state[2], self.state[3]

    def getDynamicsInfo(self, body, link):
        return (self.mass,)

    def getMatrixFromQuaternion(self, q):
        from scipy.spatial.transform import Rotation
        return tuple(Rotation.from_quat(np.asarray(q, dtype=np.float64)).as_matrix().reshape(9))

    def getEulerFromQuaternion(self, q):
        from scipy.spatial.transform import Rotation
        return tuple(Rotation.from_quat(np.asarray(q, dtype=np.float64)).as_euler("xyz"))

    # -- wrench recording
    def applyExternalForce(self, body, link, force, pos, flags):
        assert flags == self.WORLD_FRAME and link == -1
        self.calls.append(("force", [float(x) for x in force], [float(x) for x in pos]))

    def applyExternalTorque(self, body, link, torque, flags):
        assert flags == self.WORLD_FRAME and link == -1
        self.calls.append(("torque", [float(x) for x in torque], None))

    def stepSimulation(self, client=None):
        assert self.next_state is not None, "no scripted pose queued"
        self.state, self.next_state = self.next_state, None
        self.n_step += 1


def install_stand_ins():
    sys.dont_write_bytecode = True
    pb = ScriptedBullet()
    pbd = types.ModuleType("pybullet_data")
    pbd.getDataPath = lambda: "/nonexistent"
    gym = types.ModuleType("gymnasium")
    spaces = types.ModuleType("gymnasium.spaces")
    rec = {}

    class _Env:
        def reset(self, seed=None, options=None):
            rec["last_seed"] = seed

    class _Box:
        def __init__(self, low, high, shape=None, dtype=np.float32):
            self.low, self.high, self.dtype = low, high, dtype
            self.shape = tuple(shape) if shape is not None else np.shape(low)

    gym.Env, spaces.Box, gym.spaces = _Env, _Box, spaces
    sys.modules.update({"pybullet": pb, "pybullet_data": pbd, "gymnasium": gym, "gymnasium.spaces": spaces})
    spec = importlib.util.spec_from_file_location("ref_enhanced_rocket_tvc_env_step",
                                                  os.path.join(REF, "env", "enhanced_rocket_tvc_env.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[spec.name] = mod
    spec.loader.exec_module(mod)
    return mod, pb


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


def curiosity_named():
    return [("0.weight", (256, 10)), ("0.bias", (256,)), ("2.weight", (256, 256)), ("2.bias", (256,)),
            ("4.weight", (8, 256)), ("4.bias", (8,))]


def state_tuple(s13):
    s = [float(x) for x in s13]
    return (tuple(s[0:3]), tuple(s[3:7]), tuple(s[7:10]), tuple(s[10:13]))


def pd_action(s13, rng, noise):
    """Stabilising gimbal command for the hover scenarios (tau_body = (0.5 T_y, -0.5 T_x, 0), env/...:539-556)."""
    qx, qy = s13[3], s13[4]
    wx, wy = s13[10], s13[11]
    a0 = -(3.0 * 2.0 * qx + 1.0 * wx)
    a1 = +(3.0 * 2.0 * qy + 1.0 * wy)
    return np.array([a0, a1]) + noise * rng.standard_normal(2)


def run_scenario(mod, pb, eo, name, T, policy, max_episode_steps=1000, thrust=None, forced_resets=(), seed=0, thrust_fn=None):
    """thrust: constant override of env.thrust_profile (the attribute _create_enhanced_rocket sets, :463), re-applied after
    every reset; thrust_fn(t, state13): per-step override (a DR-like thrust curve), applied to the reference env and to the
    oracle's params before the step and stored per step."""
    import torch
    rec = _load("gen_sac_golden", os.path.join(HERE, "gen_sac_golden.py"))
    rng = np.random.default_rng(seed)
    env = mod.EnhancedRocketTVCEnv(config={}, max_episode_steps=max_episode_steps, enable_curiosity=True)
    vals = rec.fill_params(curiosity_named(), np.random.default_rng(CUR_SEED))
    with torch.no_grad():
        for n, p in env.curiosity_module.forward_model.named_parameters():
            p.copy_(torch.from_numpy(vals[n]))
    over = dict(contact=1, auto_reset=0, distinct_window=1000, max_episode_steps=max_episode_steps)
    if thrust is not None:
        over["thrust"] = float(thrust)
    orc = eo.OracleEnv(**over)

    def do_reset():
        obs, info = env.reset(seed=7)
        if thrust is not None:
            env.thrust_profile = float(thrust)  # the attribute _create_enhanced_rocket sets (:463); a DR-like override
        orc.reset()
        return obs, info

    out = dict(action=np.zeros((T, 2)), post=np.zeros((T, 13)), thrust_t=np.zeros(T),
               f_grav=np.zeros((T, 3)), f_thrust=np.full((T, 3), np.nan), p_thrust=np.full((T, 3), np.nan),
               f_drag=np.full((T, 3), np.nan), t_aero=np.zeros((T, 3)), n_calls=np.zeros(T, np.int32),
               obs=np.zeros((T, 10), np.float32), reward=np.zeros(T), term=np.zeros(T, np.uint8), trunc=np.zeros(T, np.uint8),
               comps=np.full((T, len(COMP_KEYS)), np.nan), reset_before=np.zeros(T, np.uint8),
               reset_obs=np.zeros((T, 10), np.float32),
               info_altitude=np.zeros(T), info_tilt_deg=np.zeros(T), info_omega=np.zeros(T), info_fuel=np.zeros(T),
               info_phase=np.zeros(T, np.int32), info_success=np.zeros(T, np.uint8), info_step=np.zeros(T, np.int32),
               info_criteria10=np.zeros(T, np.uint8), info_position=np.zeros((T, 3)))
    obs0, info0 = do_reset()
    out["reset_before"][0] = 1
    out["reset_obs"][0] = obs0
    reset_info_keys = sorted(info0.keys())
    pending = False
    step_info_keys = None
    for t in range(T):
        if pending or t in forced_resets:
            o, _ = do_reset()
            out["reset_before"][t] = 1
            out["reset_obs"][t] = o
            pending = False
        pre = orc.state13()
        a = np.asarray(policy(t, pre, rng), dtype=np.float64)
        out["action"][t] = a
        if thrust_fn is not None:
            env.thrust_profile = orc.p.thrust = float(thrust_fn(t, pre))
        out["thrust_t"][t] = env.thrust_profile
        orc.step(a)  # the oracle produces the pose the scripted body shows after stepSimulation
        post = orc.state13()
        out["post"][t] = post
        pb.calls = []
        pb.next_state = state_tuple(post)
        obs, reward, term, trunc, info = env.step(a.copy())
        # -- the wrench calls, in the order the reference issues them: gravity, [thrust], [drag], aero torque
        kinds = [c[0] for c in pb.calls]
        out["n_calls"][t] = len(kinds)
        forces = [c for c in pb.calls if c[0] == "force"]
        torques = [c for c in pb.calls if c[0] == "torque"]
        assert len(torques) == 1 and 1 <= len(forces) <= 3
        out["f_grav"][t] = forces[0][1]
        assert np.array_equal(forces[0][2], pre[0:3])  # gravity is applied at the COM (:527)
        rest = forces[1:]
        if rest and not np.allclose(rest[0][2], pre[0:3], atol=0, rtol=0):  # applied off the COM: the thrust
            out["f_thrust"][t], out["p_thrust"][t] = rest[0][1], rest[0][2]
            rest = rest[1:]
        if rest:
            assert np.array_equal(rest[0][2], pre[0:3])
            out["f_drag"][t] = rest[0][1]
        out["t_aero"][t] = torques[0][1]
        out["obs"][t] = obs
        out["reward"][t] = float(reward)
        out["term"][t], out["trunc"][t] = bool(term), bool(trunc)
        for k, key in enumerate(COMP_KEYS):
            if key in info["reward_components"]:
                out["comps"][t, k] = float(info["reward_components"][key])
        assert set(info["reward_components"]).issubset(COMP_KEYS), info["reward_components"].keys()
        out["info_altitude"][t] = info["altitude"]
        out["info_tilt_deg"][t] = info["tilt_angle_deg"]
        out["info_omega"][t] = info["angular_velocity_mag"]
        out["info_fuel"][t] = info["fuel_remaining"]
        out["info_phase"][t] = PHASES.index(info["mission_phase"])
        out["info_success"][t] = bool(info["mission_successful"])
        out["info_step"][t] = info["step"]
        out["info_criteria10"][t] = bool(info["success_criteria_met"])
        out["info_position"][t] = info["position"]
        step_info_keys = sorted(info.keys())
        assert isinstance(term, bool) and isinstance(trunc, bool) and obs.dtype == np.float32
        if term or trunc:
            pending = True
    np.savez_compressed(os.path.join(HERE, f"step_ref_{name}.npz"), max_episode_steps=np.int32(max_episode_steps), **out)
    print(f"{name}: T={T} resets={int(out['reset_before'].sum())} term={int(out['term'].sum())} trunc={int(out['trunc'].sum())} "
          f"success={int(out['info_success'].sum())} thrust-off steps={int(np.isnan(out['f_thrust'][:, 0]).sum())} "
          f"drag steps={int((~np.isnan(out['f_drag'][:, 0])).sum())} min alt={out['info_altitude'].min():.3f} "
          f"phases={sorted(set(out['info_phase'].tolist()))} reward[{out['reward'].min():.1f},{out['reward'].max():.1f}]")
    env.close()
    return reset_info_keys, step_info_keys, env


def main():
    sys.path.insert(0, ROOT)
    from oracle import envoracle as eo
    mod, pb = install_stand_ins()

    # A: shipped thrust (35 N < 2 g m): falls, touches the ground around step 34, tips over / crashes; several episodes,
    #    unclipped actions now and then (np.clip at :470), one reset in mid-flight
    def pol_a(t, s, rng):
        a = rng.uniform(-0.5, 0.5, 2)
        if t % 17 == 3:
            a = a * 4.0
        return a
    keys_r, keys_s, env = run_scenario(mod, pb, eo, "nominal", 420, pol_a, forced_resets=(150,), seed=11)

    # B: near-hover thrust with a stabilising command + strong noise: a long episode (fuel crossing at step 200, the
    #    phase chain, success criteria failing and passing), truncation at max_episode_steps = 330
    run_scenario(mod, pb, eo, "hover", 700, lambda t, s, rng: pd_action(s, rng, 0.25), max_episode_steps=330, thrust=39.5, seed=12)

    # C: exact hover, quiet command: the 100-step success window fills -> terminated by success; criteria_history survives
    #    the reset (:61), so the next episode succeeds on its first step
    run_scenario(mod, pb, eo, "success", 260, lambda t, s, rng: pd_action(s, rng, 0.004), thrust=39.24, seed=13)

    # D: altitude-hold thrust curve at 3 m (outside the success band) until the fuel runs out at step 1000 (:530-533), then
    #    the un-thrusted fall: thrust-off wrench, the LANDING -> TOUCHDOWN chain, truncation cap 1100 never reached
    run_scenario(mod, pb, eo, "burnout", 1080, lambda t, s, rng: pd_action(s, rng, 0.05), max_episode_steps=1100, seed=14,
                 thrust_fn=lambda t, s: 39.24 + 2.0 * (3.0 - s[2]) - 3.0 * s[9])

    lo, hi = env.observation_space.low, env.observation_space.high
    meta = {"constants": pb.constants, "reset_info_keys": keys_r, "step_info_keys": keys_s, "component_keys": COMP_KEYS,
            "phases": [ph.value for ph in mod.MissionPhase], "obs_low": [float(x) for x in lo], "obs_high": [float(x) for x in hi],
            "action_low": float(env.action_space.low), "action_high": float(env.action_space.high),
            "action_shape": list(env.action_space.shape), "curiosity_seed": CUR_SEED,
            "thrust_profile": 35.0, "reference_file": "env/enhanced_rocket_tvc_env.py"}
    with open(os.path.join(HERE, "step_ref_meta.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print(json.dumps(pb.constants, indent=1))


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("needs /root/reference (build container only)")
    main()
