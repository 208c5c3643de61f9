#!/usr/bin/env python3
"""Generate golden vectors for the pure-Python half of the reference env step.

Runs ONLY in the build container (needs /root/reference).  It imports the reference module
env/enhanced_rocket_tvc_env.py unmodified, after placing in-process stand-in modules for
`pybullet`, `pybullet_data` and `gymnasium` in sys.modules (they are not installed in this image and
none of the code paths exercised here touches them), and drives the reference's own
  MultiObjectiveReward.compute_reward         (env/enhanced_rocket_tvc_env.py:86-224)
  EnhancedRocketTVCEnv._update_mission_phase  (:635-657)
  EnhancedRocketTVCEnv._check_mission_success (:659-695)
  EnhancedRocketTVCEnv._check_termination     (:697-721)
in the order step() calls them (:481-510) on synthetic derived-state sequences.

Output: tests/golden/env_logic_<scenario>.npz  (inputs + the reference's outputs; data only).
"""
import importlib
import os
import sys
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def load_reference_env_module():
    sys.dont_write_bytecode = True
    pb = types.ModuleType("pybullet")
    pbd = types.ModuleType("pybullet_data")
    gym = types.ModuleType("gymnasium")
    spaces = types.ModuleType("gymnasium.spaces")

    class _Env:  # base class stand-in
        def reset(self, seed=None, options=None):
            return None

    class _Box:
        def __init__(self, *a, **k):
            pass

    gym.Env = _Env
    spaces.Box = _Box
    gym.spaces = spaces
    sys.modules["pybullet"] = pb
    sys.modules["pybullet_data"] = pbd
    sys.modules["gymnasium"] = gym
    sys.modules["gymnasium.spaces"] = spaces
    # load the module file itself (the package __init__ only adds gym registration, env/__init__.py:24)
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "ref_enhanced_rocket_tvc_env", os.path.join(REF, "env", "enhanced_rocket_tvc_env.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[spec.name] = mod  # dataclasses need the module registered
    spec.loader.exec_module(mod)
    return mod


class RefLogic:
    """The reference objects, wired as EnhancedRocketTVCEnv.__init__ (:299-309) wires them."""

    def __init__(self, mod, max_episode_steps=1000):
        self.mod = mod
        env = object.__new__(mod.EnhancedRocketTVCEnv)
        env.config = {}
        env.max_episode_steps = max_episode_steps
        env.debug = False
        env.mission_success = mod.MissionSuccess()
        env.multi_objective_reward = mod.MultiObjectiveReward({})
        env.current_phase = mod.MissionPhase.BOOST
        env.mission_successful = False
        env.phase_start_time = 0
        env.current_step = 0
        self.env = env
        self.fuel = 1.0
        self.phases = list(mod.MissionPhase)

    def reset(self):
        # what reset() does to the logic state (:386-389, :462)
        e = self.env
        e.current_phase = self.mod.MissionPhase.BOOST
        e.mission_successful = False
        e.phase_start_time = 0
        e.current_step = 0
        self.fuel = 1.0

    def step(self, sc, action):
        """sc = (altitude, tilt, omega_mag, v_h, v_z_abs, x, y); returns reference outputs."""
        e = self.env
        # fuel bookkeeping of _apply_enhanced_control (:530-533) and the step counter (:478)
        if self.fuel > 0:
            self.fuel = max(0, self.fuel - 0.001)
        e.current_step += 1
        alt, tilt, wmag, vh, vz, x, y = [float(v) for v in sc]
        # keys of _get_state_dict (:618-633)
        state = {
            'position': (x, y, alt), 'altitude': alt, 'tilt_angle': tilt,
            'angular_velocity_mag': wmag, 'horizontal_velocity': vh, 'vertical_velocity': vz,
            'fuel_remaining': self.fuel, 'mission_phase': e.current_phase.value,
            'mission_successful': e.mission_successful, 'target_altitude': 3.0,
            'crashed': alt < 0.1,
        }
        e._update_mission_phase(state)
        e._check_mission_success(state)
        reward, comps = e.multi_objective_reward.compute_reward(state, action, e.mission_success)
        term, trunc = e._check_termination(state)
        names = ['mission_completion', 'safety_compliance', 'fuel_efficiency', 'stability_bonus',
                 'control_smoothness', 'altitude_maintenance', 'crash_penalty', 'excessive_tilt',
                 'control_saturation']
        cvec = [float(comps.get(k, 0.0)) for k in names]
        return dict(reward=float(reward), comps=cvec, term=bool(term), trunc=bool(trunc),
                    phase=self.phases.index(e.current_phase), success=bool(e.mission_successful),
                    fuel=float(self.fuel), step=int(e.current_step))


def run_scenario(mod, name, scalars, actions, reset_on_done=True, max_episode_steps=1000):
    ref = RefLogic(mod, max_episode_steps)
    T = len(scalars)
    out = dict(reward=np.zeros(T), comps=np.zeros((T, 9)), term=np.zeros(T, np.uint8),
               trunc=np.zeros(T, np.uint8), phase=np.zeros(T, np.int32), success=np.zeros(T, np.uint8),
               fuel=np.zeros(T), step=np.zeros(T, np.int32), reset_before=np.zeros(T, np.uint8))
    pending_reset = False
    for t in range(T):
        if pending_reset:
            ref.reset()
            out['reset_before'][t] = 1
            pending_reset = False
        r = ref.step(scalars[t], np.array(actions[t], dtype=np.float64))
        for k in ('reward', 'term', 'trunc', 'phase', 'success', 'fuel', 'step'):
            out[k][t] = r[k]
        out['comps'][t] = r['comps']
        if reset_on_done and (r['term'] or r['trunc']):
            pending_reset = True
    path = os.path.join(OUT, f"env_logic_{name}.npz")
    np.savez_compressed(path, scalars=np.asarray(scalars, np.float64), actions=np.asarray(actions, np.float64),
                        max_episode_steps=np.int32(max_episode_steps), **out)
    print(f"{name}: T={T} terms={int(out['term'].sum())} truncs={int(out['trunc'].sum())} "
          f"success={int(out['success'].sum())} reward[min,max]=({out['reward'].min():.3f},{out['reward'].max():.3f}) -> {path}")


def main():
    mod = load_reference_env_module()
    rng = np.random.default_rng(20251004)

    # --- S1: random derived states, random actions (exercises every branch statistically)
    T = 3000
    sc = np.zeros((T, 7))
    sc[:, 0] = rng.uniform(-0.2, 22.0, T)                 # altitude
    sc[:, 1] = np.abs(rng.normal(0.0, 0.2, T))            # tilt
    sc[:, 2] = np.abs(rng.normal(0.0, 0.3, T))            # omega mag
    sc[:, 3] = np.abs(rng.normal(0.0, 0.6, T))            # v_h
    sc[:, 4] = np.abs(rng.normal(0.0, 2.0, T))            # |v_z|
    sc[:, 5] = rng.uniform(-60, 60, T)
    sc[:, 6] = rng.uniform(-60, 60, T)
    act = rng.uniform(-1, 1, (T, 2))
    run_scenario(mod, "random", sc, act)

    # --- S2: benign flight (few terminations): fuel crossing at step 200, phase chain, truncation
    T = 2600
    sc = np.zeros((T, 7))
    t = np.arange(T)
    sc[:, 0] = np.where(t % 1000 < 230, 8.0, np.maximum(0.3, 8.0 - 0.02 * ((t % 1000) - 230)))
    sc[:, 1] = 0.09 + 0.005 * np.sin(t * 0.1)             # tilt just above the success threshold
    sc[:, 2] = 0.05 + 0.01 * np.cos(t * 0.07)
    sc[:, 3] = 0.1
    sc[:, 4] = 0.3
    act = 0.3 * np.stack([np.sin(t * 0.05), np.cos(t * 0.03)], axis=1)
    run_scenario(mod, "benign", sc, act)

    # --- S3: success window: 99 passes, one fail, 100 passes -> success; counter survives the reset
    T = 420
    sc = np.zeros((T, 7))
    sc[:, 0] = 1.5
    sc[:, 1] = 0.01
    sc[:, 2] = 0.01
    sc[:, 3] = 0.1
    sc[:, 4] = 0.5
    sc[99, 1] = 0.2                                        # break the run at the 100th entry
    act = np.zeros((T, 2))
    act[::7, 0] = 0.25
    run_scenario(mod, "success_window", sc, act)

    # --- S4: touchdown completion path (phase chain to COMPLETE) with a short episode cap
    T = 700
    sc = np.zeros((T, 7))
    k = np.arange(T) % 350
    sc[:, 0] = np.maximum(0.3, 6.0 - 0.02 * np.maximum(0, k - 205))
    sc[:, 1] = 0.02
    sc[:, 2] = 0.15                                        # stability criterion fails -> only the touchdown path
    sc[:, 3] = 0.1
    sc[:, 4] = 0.4
    sc[k > 320, 2] = 0.05
    act = np.zeros((T, 2))
    run_scenario(mod, "touchdown", sc, act, max_episode_steps=350)

    # --- S5: anti-hacking branches: alternating crash / nominal (variance > 1e4), then constant
    #         rewards (distinct fraction collapses), no reset on done so the history keeps growing
    T = 1300
    sc = np.zeros((T, 7))
    sc[:, 0] = 3.0
    sc[:, 1] = 0.01
    sc[:, 2] = 0.01
    sc[:40:2, 0] = 0.05                                    # crashed every other step
    sc[40:60, 1] = 1.2                                     # big tilt penalty -> lower clip
    act = np.zeros((T, 2))
    act[:80] = rng.uniform(-1, 1, (80, 2))
    act[600:640] = rng.uniform(-1, 1, (40, 2))
    run_scenario(mod, "antihack", sc, act, reset_on_done=False, max_episode_steps=100000)

    # --- S6: every termination cause once, with resets
    rows = [
        (3.0, 0.01, 0.01, 0.1, 0.1, 0, 0),     # fine
        (0.05, 0.01, 0.01, 0.1, 0.1, 0, 0),    # crashed
        (3.0, 0.6, 0.01, 0.1, 0.1, 0, 0),      # tilt
        (20.5, 0.01, 0.01, 0.1, 0.1, 0, 0),    # too high
        (3.0, 0.01, 0.01, 0.1, 0.1, 40, 31),   # too far (r > 50)
        (3.0, 0.01, 0.01, 0.1, 0.1, 30, 39),   # r = 49.2, fine
        (0.1, 0.52, 0.1, 0.5, 2.0, 0, 0),      # exactly on thresholds
        (0.2, 0.087, 0.1, 0.5, 2.0, 0, 0),
        (2.0, 0.05, 0.2, 0.0, 0.0, 0, 0),
        (20.0, 0.1, 0.1, 0.0, 0.0, 50.0, 0.0),
    ]
    sc = np.array(rows * 3, dtype=np.float64)
    act = np.array([[0.0, 0.0], [0.5, 0.0], [0.3, 0.4], [0.9, 0.0], [0.63639610306789274, 0.63639610306789274],
                    [1.0, 1.0], [-1.0, 0.2], [0.0, -0.5], [0.1, 0.1], [0.0, 0.9000000000000001]] * 3)
    run_scenario(mod, "terminations", sc, act, max_episode_steps=4)


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("needs /root/reference (build container only)")
    main()
