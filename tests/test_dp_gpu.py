"""Data-parallel train loop with two ranks sharing one GPU (gloo stands in for RCCL, which refuses two ranks per device):
the replicas start from rank 0's weights, see different envs and batches, and must hold identical parameters after every
step because both gradient slices are averaged before their Adam steps -- on the two-stream schedule the bench uses."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ["TVC_ROOT"])
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=world)
from tvc_ai_amd.trainer import VecTrainer
tr = VecTrainer(256, device="cuda:0", family=int(os.environ["TVC_FAMILY"]), batch_size=64, replay_capacity=4096, seed=31 + rank,
                rank=rank, world=world, overlap=True, defer_join=os.environ.get("TVC_SEGMENTS") == "1")
if os.environ.get("TVC_SEGMENTS") == "1":   # the launch mode of bench.py at world > 1: segment graphs around the two collectives
    for _ in range(3):
        tr.step(True)
    fn = tr.capture_segments()
    for _ in range(4):
        fn()
else:
    for _ in range(7):
        tr.step(True)
torch.cuda.synchronize()
assert tr.steps == 7
p = tr.sac.params.cpu()
mine = torch.stack([p.sum(), p.abs().sum(), (p * p).sum()]).double()
both = [torch.zeros_like(mine) for _ in range(world)]
dist.all_gather(both, mine)
obs = tr.obs[tr.cur].cpu().double().sum().reshape(1)
obs_all = [torch.zeros_like(obs) for _ in range(world)]
dist.all_gather(obs_all, obs)
assert torch.equal(both[0], both[1]), (both[0].tolist(), both[1].tolist())          # identical replicas
assert tr.sync.calls == 2 * 7 and tr.env.cfg.env_id_offset == rank * 256             # two all-reduces per update; sharded env ids
assert torch.isfinite(p).all()
print("rank", rank, "ok", float(mine[1]), flush=True)
dist.destroy_process_group()
'''


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("family,segments", [(0, 0), (1, 0), (0, 1)])
def test_two_ranks_keep_identical_replicas(family, segments):
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), TVC_ROOT=ROOT,
                   TVC_FAMILY=str(family), TVC_SEGMENTS=str(segments))
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                                      text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            p.kill()
            out, _ = p.communicate()
        outs.append(out)
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {rank} failed:\n{out[-3000:]}"
        assert f"rank {rank} ok" in out
