"""The reference's hierarchical acting path on the device: HierarchicalAgent (agent/multi_algorithm_agent.py:353-417).

``config.yaml`` ships with ``hierarchical_rl.enabled: true`` (:103-104), and then the reference's ``get_action`` (:751-754)
does NOT act with the SAC policy it trains: a goal is drawn from ``softmax(high_level_policy(state))`` with
``torch.multinomial`` and a second, goal-conditioned ``TransformerPolicyNetwork(obs + 4, A, NetworkConfig())`` produces
mean / log_std.  Neither net is ever trained (their optimisers :386-387 are never stepped).  This module reproduces that
path for N rows per call: goal logits through the small-MLP handle (Linear-GELU-LN x2 + Linear), the categorical draw
and the ``[state | one_hot(goal)]`` input in one kernel (``tvc_goal_sample``), the low-level policy through the acting
kernels of the SAC handle built with ``use_se=1`` (NetworkConfig() defaults include the SqueezeExcitation block).
"""
import ctypes as C
from typing import Optional

import torch

from . import _native as nat
from .agent import NativeSAC, sac_cfg
from .curiosity import SmallMLP

GOALS = ["hover", "land", "recover", "maintain_altitude"]  # agent/...:362


class HierarchicalPolicy:
    def __init__(self, obs_dim: int = 10, action_dim: int = 2, device="cuda:0", max_rows: int = 4096, seed: int = 0,
                 pe_rows: int = 1, train_mode: bool = False, dropout_p: float = 0.1):
        """train_mode=True: the low-level policy acts with its Dropout(NetworkConfig().dropout = 0.1) active, as in the reference,
        whose nets are never put in eval mode (agent/...:377-381, 765); False = the deterministic net the goldens pin."""
        self.L = nat.load()
        self.device = torch.device(device)
        self.obs_dim, self.action_dim, self.goal_dim = obs_dim, action_dim, len(GOALS)
        self.max_rows = max_rows
        # high-level policy (:366-374): default torch init, never trained
        self.high = SmallMLP([obs_dim, 256, 128, self.goal_dim], device=self.device, max_rows=max_rows, act=1, seed=seed,
                             layernorm=True)
        # low-level policy (:377-381): TransformerPolicyNetwork(obs + goals, A, NetworkConfig()); acting only
        self.low = NativeSAC(sac_cfg(0, obs_dim=obs_dim + self.goal_dim, act_dim=action_dim, use_se=1, batch_size=1,
                                     max_act_rows=max_rows, pe_rows=pe_rows,
                                     dropout_p=dropout_p if train_mode else 0.0), device=self.device, seed=seed + 1)
        self.train_mode = bool(train_mode) and dropout_p > 0.0
        g = torch.Generator().manual_seed(seed + 2)
        self.goal_embedding = torch.randn(self.goal_dim, 32, generator=g)  # nn.Embedding(4, 32) (:384), unused by the reference
        self._gen = torch.Generator(device=self.device).manual_seed(seed + 3)
        self._state_goal = torch.empty((max_rows, obs_dim + self.goal_dim), dtype=torch.float32, device=self.device)
        self._goal_idx = torch.empty((max_rows,), dtype=torch.int32, device=self.device)

    def close(self):
        self.high.close()
        self.low.close()

    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def goal_logits(self, state: torch.Tensor) -> torch.Tensor:
        return self.high.forward(state)

    def select_goal(self, state: torch.Tensor, u: Optional[torch.Tensor] = None) -> torch.Tensor:
        """-> int32 goal index per row; u = uniform [0,1) draws (default: from this object's device generator)"""
        n = state.shape[0]
        self._prepare(state, u)
        return self._goal_idx[:n].clone()

    def _prepare(self, state, u):
        n = state.shape[0]
        if n > self.max_rows:
            raise nat.TvcError(f"{n} rows > max_rows={self.max_rows}")
        if u is None:
            u = torch.rand((n,), device=self.device, generator=self._gen)
        logits = self.high.forward(state)
        nat.check(self.L.tvc_goal_sample(logits.data_ptr(), u.data_ptr(), state.data_ptr(), state.stride(0), self.obs_dim,
                                         self.goal_dim, n, self._state_goal.data_ptr(), self._goal_idx.data_ptr(), self._stream()))
        return self._state_goal[:n]

    def get_action(self, state: torch.Tensor, goal_idx: torch.Tensor):
        """-> (mean, log_std, None) for the given goals, like HierarchicalAgent.get_action (the value head is dead compute)"""
        onehot = torch.nn.functional.one_hot(goal_idx.long(), self.goal_dim).to(torch.float32)
        sg = torch.cat([state, onehot], dim=-1).contiguous()
        _, mean, ls = self.low.act(sg, None, clamp=False, train_mode=self.train_mode)
        return mean, ls, None

    def act(self, state: torch.Tensor, eps: Optional[torch.Tensor] = None, u: Optional[torch.Tensor] = None, clamp: bool = True,
            share_rows: int = 0, x3: bool = False):
        """select_goal + get_action + Normal sample in four launches' worth of host calls: -> (action, mean, log_std, goal_idx).
        share_rows = k > 0: rows [0, k) go through the one-launch acting kernel in its CU-sharing form (tvc_sac_act flags bit 2),
        the rest in its whole-chip form, as VecTrainer does for the SAC policy.  x3 = True: the low-level policy's one-launch kernel on the
        bf16 matrix pipe with split operands (fp32-exact, tvc_sac_act flags bit 4; >= 16 384 rows)."""
        sg = self._prepare(state.contiguous(), u)
        n = state.shape[0]
        if 0 < share_rows < n and not self.train_mode:
            out = tuple(torch.empty((n, self.action_dim), dtype=torch.float32, device=self.device) for _ in range(3))
            for lo, hi, sh in ((0, share_rows, True), (share_rows, n, False)):
                self.low.act(sg[lo:hi], None if eps is None else eps[lo:hi], out=tuple(o[lo:hi] for o in out), clamp=clamp, share_cus=sh,
                             x3=x3)
            act, mean, ls = out
        else:
            act, mean, ls = self.low.act(sg, eps, clamp=clamp, train_mode=self.train_mode, share_cus=share_rows >= n > 0,
                                         x3=x3 and not self.train_mode)
        return act, mean, ls, self._goal_idx[:n]
