"""N>1 path on CPU: world_size-2 gloo processes exercise the data-parallel plumbing (env sharding, flat gradient
all-reduce + scale, parameter broadcast) that the GPU ranks use with RCCL."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tvc_ai_amd.parallel import GradSync, broadcast_parameters, max_over_ranks, shard_range


def test_shard_range_partitions_exactly():
    for total in (65536, 32768, 4096, 1000, 7):
        for world in (1, 2, 3, 8):
            spans = [shard_range(r, world, total) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            for a, b in zip(spans, spans[1:]):
                assert a[1] == b[0]
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(2, 2, 10)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(100 + rank)
        params = torch.randn(1000)
        broadcast_parameters(params)           # everyone now holds rank 0's weights
        grads = torch.arange(2000, dtype=torch.float32) * (rank + 1)
        sync = GradSync()
        n_policy = 1500
        sync(grads[n_policy:])                 # critic slice, then actor slice: two calls per update
        sync(grads[:n_policy])
        # replica update with the folded 1/world scale (what the Adam kernel does with grad_scale)
        new_params = params - 0.1 * (grads[:1000] * sync.grad_scale)
        mx = max_over_ranks(float(rank + 1), torch.device("cpu"))
        q.put((rank, params[:5].tolist(), grads[[0, 1499, 1500, 1999]].tolist(), new_params.sum().item(), sync.calls,
               sync.bytes, mx))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_gradient_sync():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, p0, g0, s0, c0, b0, m0), (r1, p1, g1, s1, c1, b1, m1) = res
    assert p0 == p1                                   # broadcast made the replicas identical
    assert g0 == g1 == [0.0, 1499 * 3.0, 1500 * 3.0, 1999 * 3.0]   # sum over ranks of i*(rank+1)
    assert abs(s0 - s1) < 1e-3                        # identical update on every replica
    assert c0 == c1 == 2 and b0 == b1 == 2000 * 4
    assert m0 == m1 == 2.0


def test_bench_refuses_to_run_fewer_ranks_than_asked():
    """`python bench.py --gpus 8` run directly used to execute on ONE GPU and print n_gpus: 1.  Now a direct
    `--gpus N` either launches its own N ranks or, when fewer than N GPUs are visible, exits non-zero before touching a GPU."""
    import subprocess
    import sys
    if torch.cuda.device_count() >= 8:
        pytest.skip("an 8-GPU node launches the ranks instead of refusing")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "TVC_FORCE_DEVICE")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 2, (p.returncode, p.stderr[-500:])
    assert "refusing to run fewer ranks" in p.stderr and p.stdout.strip() == ""


def test_bench_defaults_to_the_reference_reward_window_for_the_train_workload():
    """the headline train line runs MultiObjectiveReward's diversity test over the reference's whole 1000-entry deque (env/...:221);
    the 10-entry approximation is opt-in (--reward-window 10), physics-only lines keep it (they measure the W10 kernel)"""
    import importlib.util
    from types import SimpleNamespace
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    assert b.reward_window(SimpleNamespace(reward_window=0, exact_reward=False), "train") == 1000
    assert b.reward_window(SimpleNamespace(reward_window=10, exact_reward=False), "train") == 10
    assert b.reward_window(SimpleNamespace(reward_window=0, exact_reward=False), "physics") == 10
    assert b.reward_window(SimpleNamespace(reward_window=10, exact_reward=True), "physics") == 1000
