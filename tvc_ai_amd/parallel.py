"""Data-parallel plumbing (K10): env sharding by rank and one flat gradient all-reduce per optimiser step.

The reference has no distributed code at all; BASELINE.json asks for env batches sharded over the
8 GPUs of a node with an RCCL all-reduce of the critic gradients over xGMI.  Design: rank r owns
envs [r*N/R, (r+1)*N/R), its own replay buffer and Philox streams keyed by GLOBAL env id; weights start
identical (broadcast from rank 0) and stay identical because critic AND actor gradients are averaged before
each Adam step.  Messages are 0.5-9 MB, i.e. latency-bound: one call per phase over the flat buffer.
Backend-agnostic (nccl == RCCL on ROCm; gloo in the CPU tests).
"""
from typing import Tuple

import torch
import torch.distributed as dist


def shard_range(rank: int, world: int, total: int) -> Tuple[int, int]:
    """[start, stop) of the env ids owned by `rank` (total need not divide evenly)."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world {world}")
    base, rem = divmod(total, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


class GradSync:
    """all_reduce(sum) of a flat gradient slice; the division by world size is folded into the Adam kernel
    (grad_scale), so no extra pass over the buffer is needed."""

    def __init__(self, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.grad_scale = 1.0 / self.world
        self.calls = 0
        self.bytes = 0

    def __call__(self, flat: torch.Tensor):
        if self.world > 1:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        self.calls += 1
        self.bytes += flat.numel() * flat.element_size()
        return flat


def broadcast_parameters(flat_params: torch.Tensor, src: int = 0, group=None):
    """make every replica start from rank `src`'s weights"""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(flat_params, src=src, group=group)
    return flat_params


def max_over_ranks(x: float, device) -> float:
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return x
    t = torch.tensor([x], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
