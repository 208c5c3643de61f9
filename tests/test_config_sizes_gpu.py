"""BASELINE.json configs at their STATED sizes (the round-1 trainer tests ran 256-512 envs, batch 64, replay <= 5000):

  configs[2]  4 096 envs + SAC, replay 1 000 000 rows, batch 256, reference network shapes -- full train loop
  configs[4]  65 536 envs, full domain randomisation (config.yaml:340-349) at curriculum stage 5 + the curriculum driver attached
              and advancing a stage in mid-run

No oracle replay at these sizes: size-independent properties (finite, unit quaternions, reward clip range, replay ring
bookkeeping, device-side episode statistics equal to a host-side count, DR draws inside the active stage's ranges).
"""
import numpy as np
import pytest
import torch

from tests import parity_log

pytestmark = pytest.mark.gpu


def test_config2_4096_envs_sac_batch256_replay_1m():
    from tvc_ai_amd.trainer import VecTrainer
    n, steps = 4096, 60
    tr = VecTrainer(n, family=0, batch_size=256, replay_capacity=1_000_000, seed=42, overlap=True)
    assert tr.dropout_p == 0.1  # the reference's update runs in train mode
    p0 = tr.sac.params.clone()
    losses = []
    for k in range(steps):
        tr.step(True)
        if k % 10 == 9:
            losses.append(tr.sac.losses.cpu().tolist())
    torch.cuda.synchronize()
    assert len(tr.rb) == steps * n == 245_760                       # below capacity: no wrap yet
    rows, meta = tr.rb.export()
    assert meta == [(steps * n) % 1_000_000, steps * n, steps]       # head, size, batches drawn (one per step)
    assert torch.isfinite(rows).all()
    s, a, r, s2, d = rows[:, :10], rows[:, 10:12], rows[:, 12], rows[:, 13:23], rows[:, 23]
    assert (a.abs() <= 1.0).all() and r.max() <= 200.0 and r.min() >= -1000.0
    assert ((d == 0) | (d == 1)).all() and 0.0 < d.mean() < 0.2      # episodes do end; most transitions are not terminal
    assert ((s[:, :4].norm(dim=1) - 1).abs() < 1e-4).all() and ((s2[:, :4].norm(dim=1) - 1).abs() < 1e-4).all()
    assert np.all(np.isfinite(losses)), losses
    assert torch.isfinite(tr.sac.params).all() and not torch.equal(p0, tr.sac.params)
    assert tr.sac.adam_steps() == [steps, steps]
    # wrap the 1 M ring: 200 more collect-only steps = 1 064 960 rows in total
    for _ in range(200):
        tr.step(False)
    torch.cuda.synchronize()
    assert len(tr.rb) == 1_000_000
    _, meta = tr.rb.export()
    assert meta[0] == ((steps + 200) * n) % 1_000_000
    # the update still runs on a wrapped buffer
    tr.step(True)
    torch.cuda.synchronize()
    assert np.all(np.isfinite(tr.sac.losses.cpu().numpy()))
    parity_log.record("config2_4096_envs_b256_replay1m", envs=n, steps=steps, replay_rows=1_000_000, losses_last=losses[-1],
                      done_fraction=float(d.mean()))
    tr.close()


def test_config4_65536_envs_full_dr_stage5_with_curriculum_driver():
    from tvc_ai_amd.curriculum import CurriculumDriver
    from tvc_ai_amd.env import CURRICULUM_STAGES, default_curriculum_config, dr_from_yaml
    from tvc_ai_amd.trainer import VecTrainer
    n = 65536
    dr5 = dr_from_yaml({}, 5)
    assert dr5 == dict(dr_enabled=1, dr_mass_var=0.3, dr_thrust_std=0.2, dr_cg_max=0.1, dr_wind_std=3.0, dr_obs_noise_std=0.02,
                       dr_init_tilt_max=0.7)
    tr = VecTrainer(n, family=0, batch_size=256, replay_capacity=1_000_000, seed=42, overlap=True, **dr_from_yaml({}, 4))
    drv = CurriculumDriver(default_curriculum_config())
    assert [s.name for s in drv.stages] == [s["name"] for s in CURRICULUM_STAGES[1:]]
    drv.current_stage_idx = 3                                   # "advanced_control" = stage 4 of the YAML
    # make THIS stage's advancement rule reachable by an untrained policy (the last stage keeps the YAML's 0.9 success rate)
    drv.stages[3].success_criteria = {"min_success_rate": 0.0, "min_avg_reward": -1e9, "evaluation_episodes": 50}
    drv.current_step = 0
    tr.attach_curriculum(drv, every=10, min_episodes=50)
    assert abs(tr.env.cfg.dr_wind_std - 2.0) < 1e-12 and abs(tr.env.cfg.dr_init_tilt_max - 0.4) < 1e-12
    host_eps = 0
    stage_seen = [drv.current_stage_idx]
    for k in range(80):
        tr.step(True)
        host_eps += int((tr.env.term | tr.env.trunc).sum().item())   # host-side count of finished episodes (synchronises)
        stage_seen.append(drv.current_stage_idx)
        o = tr.obs[tr.cur]
        assert torch.isfinite(o).all() and torch.isfinite(tr.env.rew).all()
        assert tr.env.rew.max() <= 200.0 + 1.0 and tr.env.rew.min() >= -1000.0   # +1: curiosity off here, clip is exact
    torch.cuda.synchronize()
    st = tr.env.episode_stats()
    assert st["episodes"] == host_eps and st["episodes"] > n        # the device-side statistics count what the host counts
    assert 0 <= st["successes"] <= st["episodes"] and st["length_sum"] >= st["episodes"]
    assert 3 in stage_seen and stage_seen[-1] == 4, stage_seen       # the driver advanced to "extreme_robustness" in mid-run
    assert tr.curriculum_log[0]["stage_before"] == 3 and tr.curriculum_log[0]["stage_after"] == 3   # < half of the stage elapsed
    assert any(e["stage_before"] == 3 and e["stage_after"] == 4 for e in tr.curriculum_log), tr.curriculum_log
    assert abs(tr.env.cfg.dr_wind_std - 3.0) < 1e-12 and abs(tr.env.cfg.dr_mass_var - 0.3) < 1e-12 \
        and abs(tr.env.cfg.dr_init_tilt_max - 0.7) < 1e-12          # stage 5 ranges are live in the env handle
    # envs that restarted after the change carry stage-5 draws: mass scale in 1 +- 0.3 with sd ~ 0.3 / sqrt 3, wind sd ~ 3 N
    for _ in range(120):
        tr.step(False)
    par = tr.env.export_state()["params"].cpu().numpy()
    ms, ts, cg, wx, wy = par[:, 0], par[:, 1], par[:, 2], par[:, 3], par[:, 4]
    assert ms.min() >= 0.7 - 1e-6 and ms.max() <= 1.3 + 1e-6 and 0.15 < ms.std() < 0.19
    assert ts.min() >= 0.5 and ts.max() <= 1.5 and 0.18 < ts.std() < 0.22
    assert np.abs(cg).max() <= 0.1 + 1e-6 and 2.7 < wx.std() < 3.3 and 2.7 < wy.std() < 3.3
    q = tr.env.export_state()["dyn"][:, 3:7]
    assert ((q.norm(dim=1) - 1).abs() < 1e-5).all()
    assert np.all(np.isfinite(tr.sac.losses.cpu().numpy())) and torch.isfinite(tr.sac.params).all()
    parity_log.record("config4_65536_envs_dr_stage5_curriculum", envs=n, episodes=st["episodes"], successes=st["successes"],
                      mean_return=st["return_sum"] / st["episodes"], mean_length=st["length_sum"] / st["episodes"],
                      curriculum_log=tr.curriculum_log[:4], mass_scale_sd=float(ms.std()), wind_sd=float(wx.std()))
    tr.close()


def test_episode_statistics_match_a_host_side_replay():
    """tvc_env_set_episode_stats against the same bookkeeping done on the host from the step outputs (small N, exact)."""
    from tvc_ai_amd import VecRocketTVCEnv
    n = 777
    env = VecRocketTVCEnv(n, device="cuda:0", seed=5, max_episode_steps=40)
    env.enable_episode_stats()
    env.reset()
    g = torch.Generator(device="cuda").manual_seed(1)
    ret = np.zeros(n)
    length = np.zeros(n)
    eps = rsum = lsum = 0.0
    for t in range(150):
        a = torch.rand((n, 2), device="cuda", generator=g) * 0.6 - 0.3
        _, rew, term, trunc, _ = env.step(a)
        r, done = rew.cpu().numpy().astype(np.float64), (term | trunc).cpu().numpy().astype(bool)
        ret += r
        length += 1
        eps += done.sum()
        rsum += ret[done].sum()
        lsum += length[done].sum()
        ret[done] = 0
        length[done] = 0
    st = env.episode_stats()
    assert st["episodes"] == eps and st["length_sum"] == lsum and eps > n
    assert abs(st["return_sum"] - rsum) <= 1e-4 * abs(rsum)  # fp32 running returns on the device vs fp64 on the host
    env.close()
