"""Host-side mirror of the reference env surface over libtvc_hip.so.

* ``VecRocketTVCEnv``      -- N envs on one MI355X, device tensors in/out (the fast path).
* ``EnhancedRocketTVCEnv`` -- same constructor / reset() / step() / close() surface and return
  types as the reference class (env/enhanced_rocket_tvc_env.py:271-753) for N = 1, so a loop written
  like scripts/train.py:535-620 runs unchanged.

The arithmetic runs in hand-written HIP kernels (csrc/tvc_env.hip); nothing here computes physics.
"""
import ctypes as C
from dataclasses import dataclass
from enum import Enum
from typing import Optional

import numpy as np
import torch

from . import _native as nat

PHASE_NAMES = ["boost", "coast", "landing", "touchdown", "hover", "complete", "failed"]  # ref :21-29


class MissionPhase(Enum):
    """ref env/enhanced_rocket_tvc_env.py:21-29; member ORDER is the phase index of obs[8] = index / 7 (:593) and of
    the kernel (tvc_env_device.h: phase_value)."""
    BOOST = "boost"
    COAST = "coast"
    LANDING = "landing"
    TOUCHDOWN = "touchdown"
    HOVER = "hover"
    COMPLETE = "complete"
    FAILED = "failed"


class SuccessCriteria(Enum):
    """ref env/enhanced_rocket_tvc_env.py:31-37"""
    ATTITUDE = "attitude"
    VELOCITY = "velocity"
    POSITION = "position"
    STABILITY = "stability"
    FUEL = "fuel"


@dataclass
class MissionSuccess:
    """Thresholds of the reference's success detector (ref :39-56).  The kernel hard-codes the same numbers
    (tvc_env_device.h: epilogue); this object is the read-only surface scripts look at."""
    max_tilt_angle: float = 0.087
    max_angular_velocity: float = 0.1
    max_horizontal_velocity: float = 0.5
    max_vertical_velocity: float = 2.0
    min_altitude: float = 0.2
    max_altitude: float = 2.0
    position_tolerance: float = 1.0
    success_duration: int = 100
    boost_duration: int = 100
    coast_duration: int = 200
    landing_duration: int = 300
    touchdown_duration: int = 100


# key order of info['reward_components'] (MultiObjectiveReward.compute_reward, ref :97-112, penalties :189-207)
REWARD_COMPONENT_KEYS = ["mission_completion", "safety_compliance", "fuel_efficiency", "stability_bonus",
                         "control_smoothness", "altitude_maintenance", "crash_penalty", "excessive_tilt",
                         "control_saturation"]
OBS_DIM = 10
ACT_DIM = 2
EP_SLOTS = 256  # TVC_EP_SLOTS of include/tvc_native.h

# curriculum stage table = config/config.yaml:236-286 (stage 0 = the shipped, un-randomised env)
CURRICULUM_STAGES = [
    dict(name="nominal", wind_force=0.0, mass_variation=0.0, initial_tilt_max=0.0, success_threshold=0.0),
    dict(name="hover_training", wind_force=0.0, mass_variation=0.05, initial_tilt_max=0.05, success_threshold=0.7),
    dict(name="disturbance_rejection", wind_force=0.5, mass_variation=0.1, initial_tilt_max=0.1, success_threshold=0.75),
    dict(name="moderate_control", wind_force=1.0, mass_variation=0.15, initial_tilt_max=0.2, success_threshold=0.8),
    dict(name="advanced_control", wind_force=2.0, mass_variation=0.2, initial_tilt_max=0.4, success_threshold=0.85),
    dict(name="extreme_robustness", wind_force=3.0, mass_variation=0.3, initial_tilt_max=0.7, success_threshold=0.9),
]


class Box:
    """Minimal stand-in for gymnasium.spaces.Box (gymnasium is not installed in this image)."""

    def __init__(self, low, high, shape=None, dtype=np.float32, seed=None):
        self.dtype = np.dtype(dtype)
        if shape is None:
            shape = np.shape(low)
        self.shape = tuple(shape)
        self.low = np.broadcast_to(np.asarray(low, dtype=self.dtype), self.shape).copy()
        self.high = np.broadcast_to(np.asarray(high, dtype=self.dtype), self.shape).copy()
        self._rng = np.random.default_rng(seed)

    def sample(self):
        return self._rng.uniform(self.low, self.high).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

    def seed(self, seed=None):
        self._rng = np.random.default_rng(seed)

    def __repr__(self):
        return f"Box({self.low.min()}, {self.high.max()}, {self.shape}, {self.dtype})"


def _spaces():
    # ref :354-379 (_setup_spaces); bounds are declared, not enforced, exactly like the reference
    lo = np.array([-1, -1, -1, -1, -10, -10, -10, 0, 0, 0], dtype=np.float32)
    hi = np.array([1, 1, 1, 1, 10, 10, 10, 1, 1, 1], dtype=np.float32)
    return Box(lo, hi, dtype=np.float32), Box(-1.0, 1.0, shape=(ACT_DIM,), dtype=np.float32)


# keys of the YAML's `tvc_native:` section that belong to agent.MultiAlgorithmAgent, not to the env
AGENT_NATIVE_KEYS = ("batch_size", "max_act_rows", "family", "pe_rows", "dropout", "acting_dropout")


def make_cfg(config: Optional[dict] = None, max_episode_steps: int = 1000, **over) -> nat.EnvCfg:
    """tvc_env_cfg from the reference YAML dict (keys under ``env:``) plus explicit overrides."""
    L = nat.load()
    cfg = nat.EnvCfg()
    L.tvc_env_default_cfg(C.byref(cfg))
    cfg.max_episode_steps = int(max_episode_steps)
    config = config or {}
    native = config.get("tvc_native", {}) if isinstance(config, dict) else {}
    native = {k: v for k, v in (native or {}).items() if k not in AGENT_NATIVE_KEYS}  # the section is shared with the agent
    for k, v in {**native, **over}.items():
        if k in ("init_pos", "init_quat"):
            arr = getattr(cfg, k)
            for i, x in enumerate(v):
                arr[i] = float(x)
        elif k == "mass_scale":
            s = float(v)
            cfg.mass *= s
            cfg.inertia_xx *= s
            cfg.inertia_zz *= s
        elif hasattr(cfg, k):
            setattr(cfg, k, type(getattr(cfg, k))(v))
        else:
            raise KeyError(f"unknown tvc_env_cfg field {k!r}")
    return cfg


STAGE_EPISODES = [0, 200, 300, 400, 500, 600]  # config/config.yaml:240-281 (`episodes:` of each stage; 0 = the build's nominal stage)


def default_curriculum_config() -> dict:
    """The `curriculum:` section of the shipped config/config.yaml:226-286 rebuilt from CURRICULUM_STAGES (stages 1-5), in the
    shape scripts/curriculum_manager.py:60-95 (and curriculum.CurriculumDriver) parses."""
    stages = {}
    for i, st in enumerate(CURRICULUM_STAGES[1:], start=1):
        stages[f"stage_{i}"] = {"name": st["name"], "episodes": STAGE_EPISODES[i],
                                "environment": {"wind_force": st["wind_force"], "mass_variation": st["mass_variation"],
                                                "initial_tilt_max": st["initial_tilt_max"],
                                                "success_threshold": st["success_threshold"]}}
    return {"enabled": True, "type": "adaptive", "stages": stages}


def dr_from_yaml(config: dict, stage: Optional[int] = None) -> dict:
    """Domain-randomisation ranges from config/config.yaml:340-349, optionally narrowed to a
    curriculum stage (config.yaml:236-286)."""
    par = (((config or {}).get("env", {}) or {}).get("domain_randomization", {}) or {}).get("parameters", {}) or {}
    out = dict(
        dr_enabled=1,
        dr_mass_var=float(par.get("mass", {}).get("variation", 0.3)),
        dr_thrust_std=float(par.get("thrust", {}).get("variation", 0.2)),
        dr_cg_max=float(par.get("cg_offset", {}).get("max", 0.1)),
        dr_wind_std=float(par.get("wind", {}).get("max_force", 3.0)),
        dr_obs_noise_std=float(par.get("sensor_noise", {}).get("std", 0.02)),
        dr_init_tilt_max=0.0,
    )
    if stage is not None:
        st = CURRICULUM_STAGES[stage]
        out["dr_mass_var"] = st["mass_variation"]
        out["dr_wind_std"] = st["wind_force"]
        out["dr_init_tilt_max"] = st["initial_tilt_max"]
        if stage == 0:
            out.update(dr_enabled=0, dr_thrust_std=0.0, dr_cg_max=0.0, dr_obs_noise_std=0.0)
    return out


class VecRocketTVCEnv:
    """N independent rocket-TVC envs stepped by one HIP kernel launch.

    reset() -> (obs[N,10], info) ; step(actions[N,2]) -> (obs, reward[N], terminated[N], truncated[N], info)
    All tensors live on ``device``.  With auto_reset (default) a finished env restarts inside the same
    step: ``obs`` then holds the first observation of its next episode and
    ``info['final_observation']`` (when requested) the terminal one; reward/flags belong to the finished
    episode.
    """

    def __init__(self, num_envs: int, device="cuda:0", config: Optional[dict] = None, max_episode_steps: int = 1000,
                 seed: int = 42, env_id_offset: int = 0, want_final_obs: bool = False, **cfg_over):
        self.L = nat.load()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise nat.TvcError("VecRocketTVCEnv needs a GPU device: there is no CPU fallback")
        self.num_envs = int(num_envs)
        self.max_episode_steps = int(max_episode_steps)
        self.cfg = make_cfg(config, max_episode_steps, seed=seed, env_id_offset=env_id_offset, **cfg_over)
        self._h = C.c_void_p()
        dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        nat.check(self.L.tvc_env_create(C.byref(self.cfg), self.num_envs, dev_index, C.byref(self._h)))
        n = self.num_envs
        self.obs = torch.empty((n, OBS_DIM), dtype=torch.float32, device=self.device)
        self.rew = torch.empty((n,), dtype=torch.float32, device=self.device)
        self.term = torch.empty((n,), dtype=torch.uint8, device=self.device)
        self.trunc = torch.empty((n,), dtype=torch.uint8, device=self.device)
        self.final_obs = torch.empty((n, OBS_DIM), dtype=torch.float32, device=self.device) if want_final_obs else None
        self.reward_components = None
        self._ep_ret = self._ep_sums = None  # device-side episode statistics (enable_episode_stats)
        self.observation_space, self.action_space = _spaces()

    # -- lifecycle
    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self.L.tvc_env_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    # -- Gymnasium-vector style surface
    def reset(self, seed: Optional[int] = None, options: Optional[dict] = None, mask: Optional[torch.Tensor] = None,
              hard: bool = False):
        if mask is not None:
            mask = mask.to(device=self.device, dtype=torch.uint8).contiguous()
        nat.check(self.L.tvc_env_reset(self._h, nat.ptr(mask), 1 if hard else 0, self.obs.data_ptr(), self._stream()))
        if self._ep_ret is not None:  # a reset env starts a new episode: its running return starts at zero
            if mask is None:
                self._ep_ret.zero_()
            else:
                self._ep_ret.masked_fill_(mask.bool(), 0.0)
        return self.obs, {}

    def step(self, actions: torch.Tensor, out_obs: Optional[torch.Tensor] = None):
        """out_obs: optional [N,10] tensor that receives the observation instead of self.obs (lets a caller
        keep the previous observation alive without a copy)."""
        obs = self.obs if out_obs is None else out_obs
        if actions.shape != (self.num_envs, ACT_DIM):
            raise ValueError(f"actions must be [{self.num_envs},{ACT_DIM}], got {tuple(actions.shape)}")
        if actions.dtype != torch.float32 or not actions.is_contiguous() or actions.device != self.device:
            actions = actions.to(device=self.device, dtype=torch.float32).contiguous()
        nat.check(self.L.tvc_env_step(self._h, actions.data_ptr(), obs.data_ptr(), self.rew.data_ptr(),
                                      self.term.data_ptr(), self.trunc.data_ptr(), nat.ptr(self.final_obs),
                                      self._stream()))
        info = {"final_observation": self.final_obs} if self.final_obs is not None else {}
        return obs, self.rew, self.term, self.trunc, info

    def enable_episode_stats(self, on: bool = True):
        """The step kernel accumulates per-episode return / length / success on the device (tvc_env_set_episode_stats):
        what scripts/train.py:594-616 keeps per episode on the host.  Read with episode_stats()."""
        if on:
            self._ep_ret = torch.zeros((self.num_envs,), dtype=torch.float32, device=self.device)
            self._ep_sums = torch.zeros((EP_SLOTS * 16,), dtype=torch.float64, device=self.device)
            nat.check(self.L.tvc_env_set_episode_stats(self._h, self._ep_ret.data_ptr(), self._ep_sums.data_ptr()))
        else:
            nat.check(self.L.tvc_env_set_episode_stats(self._h, None, None))
            self._ep_ret = self._ep_sums = None

    def episode_stats_tensor(self):
        """device tensor [4] = episodes finished, successes, return sum, length sum (running totals since enable): the sum of the
        TVC_EP_SLOTS partial records the step kernel adds into"""
        if self._ep_sums is None:
            raise nat.TvcError("episode statistics are off: call enable_episode_stats() first")
        return self._ep_sums.view(EP_SLOTS, 16)[:, :4].sum(0)

    def episode_stats_state(self) -> dict:
        """running returns + totals for a checkpoint (VecTrainer.state_dict)"""
        if self._ep_sums is None:
            return {}
        return {"ep_ret": self._ep_ret.cpu(), "ep_sums": self.episode_stats_tensor().cpu()}

    def load_episode_stats_state(self, sd: dict):
        if not sd:
            return
        if self._ep_sums is None:
            self.enable_episode_stats()
        self._ep_ret.copy_(sd["ep_ret"])
        self._ep_sums.zero_()
        self._ep_sums[:4].copy_(sd["ep_sums"])  # the totals go into record 0

    def episode_stats(self) -> dict:
        e, s, r, l = self.episode_stats_tensor().cpu().tolist()  # synchronises
        return {"episodes": int(e), "successes": int(s), "return_sum": r, "length_sum": l}

    def enable_reward_components(self, on: bool = True):
        """Every following step() also fills ``self.reward_components`` [N,12]: the nine reward_components of the reference
        (ref :97-112, key order REWARD_COMPONENT_KEYS; a penalty that did not fire is 0), the anti-hacking adjustment, the
        unclipped total and the penalty presence mask."""
        if on and self.reward_components is None:
            self.reward_components = torch.zeros((self.num_envs, 12), dtype=torch.float32, device=self.device)
        if not on:
            self.reward_components = None
        nat.check(self.L.tvc_env_set_components_out(self._h, nat.ptr(self.reward_components)))
        return self.reward_components

    @staticmethod
    def observation_view(obs: torch.Tensor, mode: int = 10) -> torch.Tensor:
        """Column views of the 10-wide observation for the other widths the reference mentions:
        10 = shipped env (quat4, omega3, fuel, phase, progress); 8 = legacy tests / curiosity input (quat4, omega3,
        fuel); 7 = README (quat4, omega3).  No copy."""
        if mode not in (7, 8, 10):
            raise ValueError("obs mode must be 7, 8 or 10")
        return obs[:, :mode]

    def step_many(self, actions: torch.Tensor, out=None):
        """T steps in one launch with pre-supplied actions [T,N,2] (tests/benchmark.py:40-60 procedure)."""
        T = actions.shape[0]
        n = self.num_envs
        assert actions.shape == (T, n, ACT_DIM) and actions.dtype == torch.float32 and actions.is_contiguous()
        if out is None:
            out = (torch.empty((T, n, OBS_DIM), dtype=torch.float32, device=self.device),
                   torch.empty((T, n), dtype=torch.float32, device=self.device),
                   torch.empty((T, n), dtype=torch.uint8, device=self.device),
                   torch.empty((T, n), dtype=torch.uint8, device=self.device))
        obs, rew, term, trunc = out
        nat.check(self.L.tvc_env_step_many(self._h, T, actions.data_ptr(), obs.data_ptr(), rew.data_ptr(),
                                           term.data_ptr(), trunc.data_ptr(), self._stream()))
        return out

    # -- state exchange (parity tests, checkpoints)
    def export_state(self):
        n, W = self.num_envs, int(self.cfg.distinct_window)
        dyn = torch.empty((n, 13), dtype=torch.float32, device=self.device)
        aux = torch.empty((n, 8), dtype=torch.int32, device=self.device)
        pa = torch.empty((n, 2), dtype=torch.float32, device=self.device)
        par = torch.empty((n, 8), dtype=torch.float32, device=self.device)
        hist = torch.empty((n, W), dtype=torch.float32, device=self.device)
        nat.check(self.L.tvc_env_export_state(self._h, dyn.data_ptr(), aux.data_ptr(), pa.data_ptr(), par.data_ptr(),
                                              hist.data_ptr(), self._stream()))
        return dict(dyn=dyn, aux=aux, prev_action=pa, params=par, hist=hist)

    def import_state(self, dyn=None, aux=None, prev_action=None, params=None, hist=None):
        def prep(t, dtype, shape):
            if t is None:
                return None
            t = torch.as_tensor(t).to(device=self.device, dtype=dtype).contiguous()
            assert tuple(t.shape) == shape, (tuple(t.shape), shape)
            return t
        n, W = self.num_envs, int(self.cfg.distinct_window)
        dyn = prep(dyn, torch.float32, (n, 13))
        aux = prep(aux, torch.int32, (n, 8))
        pa = prep(prev_action, torch.float32, (n, 2))
        par = prep(params, torch.float32, (n, 8))
        hist = prep(hist, torch.float32, (n, W))
        nat.check(self.L.tvc_env_import_state(self._h, nat.ptr(dyn), nat.ptr(aux), nat.ptr(pa), nat.ptr(par),
                                              nat.ptr(hist), self._stream()))
        if self._ep_ret is not None and dyn is not None:  # foreign trajectories: the running returns no longer belong to them
            self._ep_ret.zero_()
        torch.cuda.current_stream(self.device).synchronize()  # keep the staging tensors alive until consumed

    def info_tensor(self):
        info = torch.empty((self.num_envs, 8), dtype=torch.float32, device=self.device)
        nat.check(self.L.tvc_env_info(self._h, info.data_ptr(), self._stream()))
        return info

    def set_domain_randomization(self, **dr):
        """Change DR ranges (e.g. on a curriculum stage change); applies from the next reset of each env.  Enqueued on the
        current stream (tvc_env_set_dr_async): steps already enqueued there keep the old ranges, later ones -- replays of a
        captured hipGraph included -- see the new ones."""
        for k, v in dr.items():
            setattr(self.cfg, k, type(getattr(self.cfg, k))(v))
        nat.check(self.L.tvc_env_set_dr_async(self._h, C.byref(self.cfg), self._stream()))

    def set_curriculum_stage(self, stage: int, config: Optional[dict] = None):
        self.set_domain_randomization(**dr_from_yaml(config or {}, stage))

    def fuel_thresholds(self):
        a, b, c = C.c_int32(), C.c_int32(), C.c_int32()
        nat.check(self.L.tvc_env_fuel_thresholds(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value


class EnhancedRocketTVCEnv:
    """Drop-in for the reference class of the same name (env/enhanced_rocket_tvc_env.py:271), N = 1.

    Same constructor keywords, ``reset() -> (np.float32[10], dict)``,
    ``step(np[2]) -> (np.float32[10], float, bool, bool, dict)``, ``close()``, ``observation_space``,
    ``action_space``, ``max_episode_steps``; info keys as scripts/train.py reads them.
    No auto-reset: the caller resets, as with the reference.  ``enable_curiosity`` adds the reference's
    untrained-forward-model bonus through VecCuriosity when that module is available.
    """

    metadata = {"render_modes": ["human", "rgb_array"], "render_fps": 60}

    def __init__(self, config: Optional[dict] = None, max_episode_steps: int = 1000, render_mode: Optional[str] = None,
                 enable_hierarchical: bool = True, enable_curiosity: bool = True, enable_physics_informed: bool = True,
                 debug: bool = False, device="cuda:0"):
        self.config = config or {}
        self.max_episode_steps = max_episode_steps
        self.render_mode = render_mode
        self.enable_hierarchical = enable_hierarchical
        self.enable_curiosity = enable_curiosity
        self.enable_physics_informed = enable_physics_informed
        self.debug = debug
        # reference semantics: whole-history distinct window, no auto-reset, no DR
        self._vec = VecRocketTVCEnv(1, device=device, config=self.config, max_episode_steps=max_episode_steps,
                                    auto_reset=0, distinct_window=1000)
        self._vec.enable_reward_components()
        self.observation_space, self.action_space = _spaces()
        self.mission_success = MissionSuccess()
        self.current_phase = MissionPhase.BOOST
        self.mission_successful = False
        self.current_step = 0
        self._act = torch.zeros((1, ACT_DIM), dtype=torch.float32, device=self._vec.device)
        self._curiosity = None
        self._prev_obs8 = None
        if enable_curiosity:
            try:
                from .curiosity import VecCuriosity
                self._curiosity = VecCuriosity(device=self._vec.device)
            except ImportError:
                self._curiosity = None

    def _info(self):
        v = self._vec.info_tensor()[0].cpu().numpy()
        return {
            "position": (float(v[0]), float(v[1]), float(v[2])),
            "altitude": float(v[2]),
            "tilt_angle_deg": float(v[3]),
            "angular_velocity_mag": float(v[4]),
            "fuel_remaining": float(v[5]),
            "mission_phase": PHASE_NAMES[int(v[6])],
            "mission_successful": False,
            "step": self.current_step,
            "success_criteria_met": bool(v[7] > 0.5),
        }

    def reset(self, seed: Optional[int] = None, options: Optional[dict] = None):
        if seed is not None:
            self.action_space.seed(seed)
        obs, _ = self._vec.reset()
        self.current_step = 0
        self.current_phase = MissionPhase.BOOST
        self.mission_successful = False
        self._prev_obs8 = None
        return obs[0].cpu().numpy().copy(), self._info()

    def step(self, action):
        a = np.clip(np.asarray(action, dtype=np.float32).reshape(ACT_DIM), -1.0, 1.0)
        self._act.copy_(torch.from_numpy(a).view(1, ACT_DIM))
        obs, rew, term, trunc, _ = self._vec.step(self._act)
        self.current_step += 1
        reward = float(rew[0].item())
        comps = self._vec.reward_components[0].cpu().numpy()
        present = int(comps[11])
        reward_components = {k: float(comps[i]) for i, k in enumerate(REWARD_COMPONENT_KEYS)
                             if i < 6 or (present >> (i - 6)) & 1}  # the dict only holds penalties that fired (ref :189-207)
        if self._curiosity is not None:
            if self._prev_obs8 is not None:  # skipped on the first step of an episode (ref :496)
                intrinsic = float(self._curiosity.intrinsic_reward(self._prev_obs8, self._act, obs[:, :8])[0].item())
                reward += intrinsic
                reward_components["curiosity"] = intrinsic
            self._prev_obs8 = obs[:, :8].clone()
        obs_np = obs[0].cpu().numpy().copy()
        terminated, truncated = bool(term[0].item()), bool(trunc[0].item())
        info = self._info()
        aux = self._vec.export_state()["aux"][0].cpu().numpy()
        self.mission_successful = bool(aux[2])
        self.current_phase = list(MissionPhase)[int(aux[1])]
        info["reward_components"] = reward_components  # ref :514-516
        info["mission_phase"] = self.current_phase.value
        info["mission_successful"] = self.mission_successful
        return obs_np, reward, terminated, truncated, info

    def render(self, mode: str = "human"):
        return None

    def close(self):
        self._vec.close()


def make_enhanced_tvc_env(**kwargs) -> EnhancedRocketTVCEnv:
    """ref: env/enhanced_rocket_tvc_env.py:756-758"""
    return EnhancedRocketTVCEnv(**kwargs)


# ref: env/__init__.py:66-102 factories (same default keyword sets)
def make_training_env(config=None, **kw):
    return EnhancedRocketTVCEnv(config=config, **{"max_episode_steps": 1000, "enable_hierarchical": True, "enable_curiosity": True,
                                                  "enable_physics_informed": True, "debug": False, **kw})


def make_evaluation_env(config=None, **kw):
    return EnhancedRocketTVCEnv(config=config, **{"max_episode_steps": 1000, "enable_hierarchical": False, "enable_curiosity": False,
                                                  "enable_physics_informed": False, "debug": False, **kw})


def make_debug_env(config=None, **kw):
    return EnhancedRocketTVCEnv(config=config, **{"render_mode": "human", "max_episode_steps": 1000, "enable_hierarchical": True,
                                                  "enable_curiosity": True, "enable_physics_informed": True, "debug": True, **kw})
