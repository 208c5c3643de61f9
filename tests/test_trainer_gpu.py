"""Vectorised train loop (act -> env step -> replay insert -> sample -> SAC update) on the GPU: both schedules
(sequential, two-stream overlapped), eager and hipGraph-captured, leave consistent state behind."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("overlap", [False, True])
def test_train_loop_schedules(overlap):
    from tvc_ai_amd.trainer import VecTrainer
    tr = VecTrainer(512, family=1, batch_size=64, replay_capacity=5000, seed=3, overlap=overlap, enable_curiosity=True)
    p0 = tr.sac.params.clone()
    for _ in range(12):
        tr.step(True)
    torch.cuda.synchronize()
    st = tr.stats()
    assert st["env_steps"] == 12 * 512 and len(tr.rb) == 5000  # 6144 rows into capacity 5000: ring wrapped
    assert np.all(np.isfinite(st["losses"])), st
    assert not torch.equal(p0, tr.sac.params) and torch.isfinite(tr.sac.params).all()
    # replicas of the target nets moved by Polyak, slowly
    n0, nc = tr.sac.n_policy, tr.sac.n_critic
    dq = (tr.sac.params[n0:n0 + 2 * nc] - p0[n0:n0 + 2 * nc]).abs().max().item()
    dt = (tr.sac.params[n0 + 2 * nc:] - p0[n0 + 2 * nc:]).abs().max().item()
    assert 0 < dt < dq
    tr.close()


def test_train_loop_in_one_hipgraph():
    from tvc_ai_amd.trainer import VecTrainer
    tr = VecTrainer(256, family=0, batch_size=64, replay_capacity=4096, seed=5, overlap=True)
    for _ in range(3):
        tr.step(True)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            for _ in range(4):
                tr.step(True)
    torch.cuda.current_stream().wait_stream(side)
    before = tr.sac.params.clone()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    assert len(tr.rb) == min(4096, 256 * (3 + 12))  # device-resident head/size advance on every replay
    assert torch.isfinite(tr.sac.params).all() and not torch.equal(before, tr.sac.params)
    aux = tr.env.export_state()["aux"].cpu().numpy()
    assert aux[:, 0].max() <= 15 + 1 and aux[:, 7].sum() > 0  # episodes ended and restarted inside the graph
    tr.close()


def test_train_loop_with_the_shipped_acting_path():
    """config.yaml defaults: the hierarchical policy acts, the safety layer corrects, SAC is what gets trained"""
    from tvc_ai_amd.trainer import VecTrainer
    tr = VecTrainer(384, family=0, batch_size=64, replay_capacity=4096, seed=9, enable_curiosity=True, enable_hierarchical=True,
                    enable_safety=True)
    p0 = tr.sac.params.clone()
    low0 = tr.hier.low.params.clone()
    for _ in range(8):
        tr.step(True)
    torch.cuda.synchronize()
    assert np.all(np.isfinite(tr.stats()["losses"]))
    assert tr.act.abs().max().item() <= 1.0 and torch.isfinite(tr.act).all()
    assert not torch.equal(p0, tr.sac.params)              # the SAC nets learn ...
    assert torch.equal(low0, tr.hier.low.params)           # ... the acting nets are never trained (as in the reference)
    tr.close()


@pytest.mark.parametrize("shipped", [False, True])
def test_checkpoint_resume_continues_the_same_run(tmp_path, shipped):
    """learner + env SoA state + observations + replay (rows, counters, Philox key) + RNG round-trip: a fresh trainer built with
    a DIFFERENT seed and loaded from the checkpoint continues like the original (equal to float-atomic summation order).
    shipped: with the curiosity / hierarchical / safety nets of the shipped config, whose weights come from the constructor seed
    and therefore must travel in the checkpoint."""
    from tvc_ai_amd.trainer import VecTrainer
    kw = dict(family=1, batch_size=64, replay_capacity=4096, overlap=True, enable_curiosity=shipped, enable_hierarchical=shipped,
              enable_safety=shipped)
    a = VecTrainer(256, seed=21, **kw)
    for _ in range(6):
        a.step(True)
    path = str(tmp_path / "resume.pt")
    a.save_checkpoint(path)
    for _ in range(5):
        a.step(True)
    torch.cuda.synchronize()
    b = VecTrainer(256, seed=99, **kw)  # different seed: everything that matters must come from the checkpoint
    b.load_checkpoint(path)
    assert b.steps == 6 and len(b.rb) == 6 * 256
    for _ in range(5):
        b.step(True)
    torch.cuda.synchronize()
    assert torch.allclose(a.sac.params, b.sac.params, atol=1e-6, rtol=0)
    ea, eb = a.env.export_state(), b.env.export_state()
    assert torch.equal(ea["aux"][:, 0], eb["aux"][:, 0])                       # per-env step counters
    assert torch.allclose(ea["dyn"], eb["dyn"], atol=1e-5)
    ra, ma = a.rb.export()
    rbb, mb = b.rb.export()
    assert ma == mb and torch.allclose(ra, rbb, atol=1e-5)
    if shipped:
        assert torch.equal(a.curiosity.params, b.curiosity.params) and torch.equal(a.hier.low.params, b.hier.low.params)
        c = VecTrainer(256, seed=5, family=1, batch_size=64, replay_capacity=4096)  # built without the acting-path nets
        with pytest.raises(ValueError, match="curiosity"):
            c.load_checkpoint(path)
        c.close()
    a.close()
    b.close()


def test_captured_step_replays_the_same_work_as_eager_steps():
    """VecTrainer.capture(): whole train steps as one hipGraph.  Replaying must do what eager steps do: same number of transitions,
    same Adam step counters, and -- same seeds, same draws from the device generator -- the same parameters up to the summation
    order of the float atomics in the backward kernels (two eager runs differ by ~1e-5 as well)."""
    from tvc_ai_amd.trainer import VecTrainer
    n, B = 2048, 64
    kw = dict(device="cuda:0", family=0, batch_size=B, replay_capacity=50_000, seed=5)

    def run(graphed):
        torch.manual_seed(123)  # the trainers draw their exploration / update noise from the default device generator
        t = VecTrainer(n, **kw)
        for _ in range(3):
            t.step(True)
        if graphed:
            replay = t.capture(steps_per_replay=2)
            for _ in range(5):
                replay()
        else:
            for _ in range(10):
                t.step(True)
        torch.cuda.synchronize()
        out = (t.steps, len(t.rb), t.sac.adam_steps(), t.sac.params.cpu().clone(), t.sac.losses.cpu().clone())
        t.close()
        return out

    sa, ra, aa, pa, la = run(True)
    sb, rb, ab, pb, lb = run(False)
    assert sa == sb == 13 and ra == rb == min(50_000, 13 * n) and aa == ab
    assert torch.isfinite(la).all() and torch.isfinite(lb).all()
    assert (pa - pb).abs().max().item() < 1e-3, (pa - pb).abs().max().item()
    torch.testing.assert_close(la, lb, rtol=1e-3, atol=1e-3)
