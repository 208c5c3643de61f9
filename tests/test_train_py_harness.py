"""scripts/train.py of the reference, UNMODIFIED, over the drop-in tree (VERDICT r2 missing item 6).

Build-container only (it reads /root/reference, which does not travel to the GPU box: the test skips there).  The reference file is
imported from where it lies -- nothing of it is copied -- with
  * sys.path = [<repo>/dropin, <repo>, /root/reference]: `env.enhanced_rocket_tvc_env` and `agent.multi_algorithm_agent`
    (scripts/train.py:44-45) resolve to dropin/, `scripts.curriculum_manager` and `utils.*` (:46-51) to the reference's own files;
  * build-authored stand-ins for the three packages the image lacks (wandb, gymnasium, seaborn: imported by train.py / its logger,
    never used on this path);
  * the DEVICE classes replaced by CPU doubles at the handle level (VecRocketTVCEnv stepping oracle/tvc_oracle.c, NativeSAC,
    VecCuriosity, SafetyLayer, HierarchicalPolicy with fixed small nets): this container has no GPU.  The host mirrors under test --
    tvc_ai_amd.env.EnhancedRocketTVCEnv and tvc_ai_amd.agent.MultiAlgorithmAgent, i.e. what a user of train.py touches -- are the
    REAL classes.  Every double is checked against the real class's public signature, so the doubles cannot drift from the product.
What it proves: StateOfTheArtTrainer.__init__ (setup_device_manager / setup_logging / setup_environment / setup_agent /
setup_training / setup_stability_manager, scripts/train.py:176-406) and run_episode (:535-620) run to completion with zero edits:
constructor keywords, attribute reads (observation_space.shape, action_space.sample), get_action / update / update_performance
calls, the batch dict with a BoolTensor `dones`, and every info key run_episode reads."""
import importlib
import inspect
import os
import sys
import types

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"

pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(REF, "scripts", "train.py")),
                                reason="needs the reference tree (build container only)")


# ------------------------------------------------------------------ CPU doubles of the device classes
class FakeVec:
    """N = 1 double of tvc_ai_amd.env.VecRocketTVCEnv over the fp64 oracle (the checker, used here as a test double)."""

    def __init__(self, num_envs, device="cpu", config=None, max_episode_steps=1000, seed=42, env_id_offset=0, want_final_obs=False,
                 **cfg_over):
        from oracle import envoracle as eo
        assert num_envs == 1
        self.device = torch.device("cpu")
        self.num_envs = 1
        self._o = eo.OracleEnv(max_episode_steps=max_episode_steps, contact=1, auto_reset=0, distinct_window=1000)
        self.reward_components = None
        self._last = None

    def reset(self, seed=None, options=None, mask=None, hard=False):
        return torch.from_numpy(self._o.reset().copy()).view(1, 10), {}

    def step(self, actions, out_obs=None):
        out = self._o.step(np.asarray(actions.reshape(-1), dtype=np.float64))
        self._last = out
        if self.reward_components is not None:
            self.reward_components[0] = torch.tensor(list(out.components), dtype=torch.float32)
        return (torch.tensor(list(out.obs), dtype=torch.float32).view(1, 10), torch.tensor([out.reward], dtype=torch.float32),
                torch.tensor([out.terminated], dtype=torch.uint8), torch.tensor([out.truncated], dtype=torch.uint8), {})

    def enable_reward_components(self, on=True):
        self.reward_components = torch.zeros((1, 12))
        return self.reward_components

    def info_tensor(self):
        e, sc = self._o.e, self._o.scalars()
        return torch.tensor([[e.pos[0], e.pos[1], e.pos[2], np.degrees(sc.tilt), sc.omega_mag, e.fuel, float(e.phase),
                              1.0 if e.success_run >= 10 else 0.0]], dtype=torch.float32)

    def export_state(self):
        e = self._o.e
        return {"aux": torch.tensor([[e.step, e.phase, e.mission_successful, e.success_run, e.hist_len, e.has_prev_action, 0,
                                      int(e.episodes)]], dtype=torch.int32)}

    def close(self):
        pass


class FakeCuriosity:
    def __init__(self, device="cpu", max_rows=4096, obs_dim=8, action_dim=2, hidden=256, seed=0):
        pass

    def intrinsic_reward(self, prev_obs, action, obs):
        return 0.01 * ((prev_obs - obs) ** 2).mean(dim=1)


class FakeSafety:
    def __init__(self, device="cpu", max_rows=4096, state_dim=10, action_dim=2, max_tilt=0.52, max_angular_velocity=5.0,
                 seed=0):
        pass

    def apply(self, state, proposed, out=None):
        return torch.clamp(proposed, -1.0, 1.0)


class FakeSAC:
    def __init__(self, cfg=None, device="cpu", seed=0, init=True, **over):
        self.cfg, self.device = cfg, torch.device("cpu")
        self.params = torch.zeros(8)
        self.updates = 0

    def act(self, obs, eps=None, out=None, clamp=True, snapshot=False, share_cus=False, train_mode=False):
        n = obs.shape[0]
        mean = 0.1 * torch.tanh(obs[:, :2])
        ls = torch.full((n, 2), -1.0)
        a = mean if eps is None else mean + torch.exp(ls) * eps
        return (torch.clamp(a, -1, 1) if clamp else a), mean, ls

    def update(self, s, a, r, s2, d, eps_next, eps_new, all_reduce=None, grad_scale=1.0):
        assert d.dtype == torch.float32 and s.shape == (1, 10) and a.shape == (1, 2)
        self.updates += 1
        self.params += 1.0
        return torch.tensor([0.5, 0.4, -0.3, 0.01])

    def close(self):
        pass


class FakeHier:
    def __init__(self, obs_dim=10, action_dim=2, device="cpu", max_rows=4096, seed=0, pe_rows=1, train_mode=False, dropout_p=0.1):
        self.calls = 0

    def act(self, state, eps=None, u=None, clamp=True, share_rows=0):
        self.calls += 1
        n = state.shape[0]
        mean = torch.zeros((n, 2))
        ls = torch.full((n, 2), -1.0)
        a = mean if eps is None else mean + torch.exp(ls) * eps
        return (torch.clamp(a, -1, 1) if clamp else a), mean, ls, torch.zeros((n,), dtype=torch.int32)

    def close(self):
        pass


def _check_double(double, real):
    """every public method of the double exists on the real class, and takes a subset of its parameter names in order"""
    for name, fn in inspect.getmembers(double, predicate=inspect.isfunction):
        if name.startswith("_") and name != "__init__":
            continue
        assert hasattr(real, name), f"{real.__name__} has no {name}"
        want = list(inspect.signature(getattr(real, name)).parameters)
        got = [p for p in inspect.signature(fn).parameters if p not in ("cfg_over", "over")]
        assert [p for p in got if p in want] == got, (real.__name__, name, got, want)


@pytest.fixture
def harness(tmp_path, monkeypatch):
    import yaml
    # stand-ins for the absent third-party packages (never exercised on this path)
    wandb = types.ModuleType("wandb")
    wandb.run = None
    wandb.init = lambda *a, **k: None
    wandb.log = lambda *a, **k: None
    wandb.finish = lambda *a, **k: None
    gym = types.ModuleType("gymnasium")
    sns = types.ModuleType("seaborn")
    for name, mod in (("wandb", wandb), ("gymnasium", gym), ("seaborn", sns)):
        monkeypatch.setitem(sys.modules, name, mod)
    for name in [m for m in sys.modules if m.split(".")[0] in ("env", "agent", "scripts", "utils")]:
        monkeypatch.delitem(sys.modules, name)
    monkeypatch.setattr(sys, "path", [os.path.join(ROOT, "dropin"), ROOT, REF] + list(sys.path))
    # device classes -> CPU doubles, default device of the two host mirrors -> cpu
    import tvc_ai_amd.agent as nagent
    import tvc_ai_amd.curiosity as ncur
    import tvc_ai_amd.env as nenv
    import tvc_ai_amd.hierarchical as nhier
    for double, real in ((FakeVec, nenv.VecRocketTVCEnv), (FakeCuriosity, ncur.VecCuriosity), (FakeSafety, ncur.SafetyLayer),
                         (FakeSAC, nagent.NativeSAC), (FakeHier, nhier.HierarchicalPolicy)):
        _check_double(double, real)
    monkeypatch.setattr(nenv, "VecRocketTVCEnv", FakeVec)
    monkeypatch.setattr(ncur, "VecCuriosity", FakeCuriosity)
    monkeypatch.setattr(ncur, "SafetyLayer", FakeSafety)
    monkeypatch.setattr(nagent, "NativeSAC", FakeSAC)
    monkeypatch.setattr(nhier, "HierarchicalPolicy", FakeHier)
    env_init, agent_init = nenv.EnhancedRocketTVCEnv.__init__, nagent.MultiAlgorithmAgent.__init__
    assert env_init.__defaults__[-1] == "cuda:0" and agent_init.__defaults__ == (None, 42)
    monkeypatch.setattr(env_init, "__defaults__", env_init.__defaults__[:-1] + ("cpu",))
    monkeypatch.setattr(agent_init, "__defaults__", (torch.device("cpu"), 42))
    # the reference's shipped YAML with the output directory moved into the test's tmp dir
    cfg = yaml.safe_load(open(os.path.join(REF, "config", "config.yaml")))
    cfg["globals"]["output_dir"] = str(tmp_path / "out")
    cfg.setdefault("logging", {}).setdefault("wandb", {})["enabled"] = False
    cfg_path = tmp_path / "config.yaml"
    cfg_path.write_text(yaml.safe_dump(cfg))
    monkeypatch.chdir(tmp_path)
    train = importlib.import_module("scripts.train")
    assert os.path.realpath(train.__file__) == os.path.realpath(os.path.join(REF, "scripts", "train.py"))
    assert sys.modules["env.enhanced_rocket_tvc_env"].__file__.startswith(os.path.join(ROOT, "dropin"))
    assert sys.modules["agent.multi_algorithm_agent"].__file__.startswith(os.path.join(ROOT, "dropin"))
    return train, str(cfg_path), cfg


def test_trainer_constructs_and_runs_an_episode_with_zero_edits(harness):
    train, cfg_path, cfg = harness
    import tvc_ai_amd.agent as nagent
    import tvc_ai_amd.env as nenv
    tr = train.StateOfTheArtTrainer(cfg_path, debug=False)                       # scripts/train.py:176-406
    assert isinstance(tr.env, nenv.EnhancedRocketTVCEnv) and isinstance(tr.eval_env, nenv.EnhancedRocketTVCEnv)
    assert isinstance(tr.agent, nagent.MultiAlgorithmAgent)
    assert tr.env.observation_space.shape == (10,) and tr.env.action_space.shape == (2,)
    assert list(tr.agent.algorithms) == ["ppo", "sac", "td3"]                    # the shipped YAML enables all three (:487-497)
    assert tr.agent.hierarchical_agent is not None and tr.agent.safety_layer is not None   # ... and both acting-path extras
    assert (tr.curriculum_manager is not None) == bool(cfg.get("curriculum", {}).get("enabled", False))
    # before the warm-up threshold (:574): acting + stepping only
    info = tr.run_episode()                                                      # scripts/train.py:535-620
    assert set(info) >= {"reward", "length", "success", "algorithm_used", "final_altitude", "final_tilt", "mission_phase",
                         "fuel_remaining", "safety_violations"}
    assert info["length"] >= 1 and np.isfinite(info["reward"]) and info["algorithm_used"] == "ppo"
    assert info["mission_phase"] in [p.value for p in nenv.MissionPhase]
    assert tr.agent.hierarchical_agent.calls == info["length"]                   # the shipped config acts through the hierarchy
    assert list(tr.agent.performance_history["ppo"]) == [info["reward"]]
    # past the threshold: every step also calls agent.update(batch) with the B = 1 batch and its BoolTensor dones (:577-585)
    tr.total_timesteps = 1000
    seen = []
    real_update = tr.agent.update

    def spy(batch, algorithm=None):
        out = real_update(batch, algorithm)
        seen.append((out, batch["dones"].dtype))
        return out
    tr.agent.update = spy
    info2 = tr.run_episode()
    assert len(seen) == info2["length"] and all(dt == torch.bool for _, dt in seen)
    assert all(isinstance(out, dict) and "error" not in out for out, _ in seen), seen[:2]
    assert set(seen[0][0]) >= {"policy_loss"}                                    # 'dynamic' selection answers 'ppo' (:693-709)
    # the same loop pinned to the accelerated learner: the SAC path of update() with the reference's own batch
    tr.agent.selection_strategy = "dynamic"
    for _ in range(3):
        tr.agent.update_performance("sac", 1e9)
    assert tr.agent.select_algorithm() == "sac"
    n0 = tr.agent.sac.updates
    info3 = tr.run_episode()
    assert tr.agent.sac.updates - n0 == info3["length"] and info3["algorithm_used"] == "sac"
    # evaluate() (:636-700) drives eval_env deterministically through the same surface
    res = tr.evaluate(episodes=1)
    assert "success_rate" in res and all(np.isfinite(v) for v in res.values())
