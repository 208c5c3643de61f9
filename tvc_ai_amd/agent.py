"""Host-side mirror of the reference agent surface over the HIP SAC learner (csrc/tvc_sac.hip).

* ``NativeSAC``           -- flat parameter / gradient / Adam buffers (torch tensors) + the C handle:
                             act(), update(), reference state_dict import/export.
* ``ReplayBuffer``        -- device-resident uniform replay (csrc/tvc_replay.hip).
* ``MultiAlgorithmAgent`` -- same constructor and methods as the reference class
                             (agent/multi_algorithm_agent.py:419-1179) for the SAC path.

Nothing here computes a forward / backward pass: torch only owns memory, RNG draws and checkpoints.
"""
import ctypes as C
import logging
import math
from collections import deque
from typing import Dict, Optional

import numpy as np
import torch

from . import _native as nat
from . import checkpoint as ckpt
from .checkpoint import positional_encoding_table  # noqa: F401  (re-exported: tests and hierarchical.py import it from here)


SacCfg = nat.SacCfg

AUTO_COUNTER = (1 << 64) - 1


def sac_cfg(family: int = 0, **over) -> SacCfg:
    cfg = SacCfg()
    nat.load().tvc_sac_default_cfg(C.byref(cfg), family)
    for k, v in over.items():
        if not hasattr(cfg, k):
            raise KeyError(f"unknown tvc_sac_cfg field {k!r}")
        setattr(cfg, k, v)
    return cfg


def dropout_seed_of(seed: int, rank: int = 0) -> int:
    """tvc_sac_cfg.dropout_seed of a learner built with `seed` on data-parallel rank `rank` (never 0: 0 pins the golden sequence)"""
    v = (int(seed) * 0x9E3779B1 + (int(rank) + 1) * 0x7F4A7C15) & 0xFFFFFFFF
    return v or 1


def tensor_table(cfg: SacCfg):
    """[(name, offset, rows, cols)] of the flat parameter buffer (host-only query)."""
    L = nat.load()
    n = L.tvc_sac_num_tensors(C.byref(cfg))
    if n < 0:
        nat.check(n)
    out = []
    name = C.create_string_buffer(128)
    off, rows, cols = C.c_int64(), C.c_int32(), C.c_int32()
    for i in range(n):
        nat.check(L.tvc_sac_tensor_info(C.byref(cfg), i, name, 128, C.byref(off), C.byref(rows), C.byref(cols)))
        out.append((name.value.decode(), off.value, rows.value, cols.value))
    return out


def _ref_to_native_name(net: str, ref_key: str):
    """reference state_dict key -> (native tensor name, row slice or None)."""
    if net == "policy":
        k = ref_key.replace("transformer_encoder.layers.", "layers.")
        if k.endswith("self_attn.in_proj_weight") or k.endswith("self_attn.in_proj_bias"):
            base = k[:k.index("self_attn")]
            return f"policy.{base}v_proj.{'weight' if k.endswith('weight') else 'bias'}", "v_rows"
        k = k.replace("self_attn.out_proj", "out_proj")
        return f"policy.{k}", None
    return f"{net}.{ref_key}", None


class NativeSAC:
    """SAC learner state on one GPU.  Parameters live in ONE flat fp32 torch tensor
    [policy | q1 | q2 | target_q1 | target_q2]; grads / Adam moments cover [policy | q1 | q2]."""

    def __init__(self, cfg: Optional[SacCfg] = None, device="cuda:0", seed: int = 0, init: bool = True, **over):
        self.L = nat.load()
        self.cfg = cfg if cfg is not None else sac_cfg(**over)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise nat.TvcError("NativeSAC needs a GPU device: there is no CPU fallback")
        if int(self.cfg.dropout_seed) == 0:  # dropout masks follow the constructor seed unless the cfg pins them
            self.cfg.dropout_seed = dropout_seed_of(seed)
        self.table = tensor_table(self.cfg)
        self.index = {n: (o, r, c) for n, o, r, c in self.table}
        self.layout = ckpt.Layout(self.table, self.cfg.family, self.cfg.d_model, self.cfg.n_layers)
        n_all = self.L.tvc_sac_param_count(C.byref(self.cfg))
        n_tr = self.L.tvc_sac_trainable_count(C.byref(self.cfg))
        self.params = torch.zeros(n_all, dtype=torch.float32, device=self.device)
        self.grads = torch.zeros(n_tr, dtype=torch.float32, device=self.device)
        self.adam_m = torch.zeros(n_tr, dtype=torch.float32, device=self.device)
        self.adam_v = torch.zeros(n_tr, dtype=torch.float32, device=self.device)
        self.losses = torch.zeros(4, dtype=torch.float32, device=self.device)
        self.n_policy = self.index["q1." + self._critic_first()][0]
        self.n_critic = (n_tr - self.n_policy) // 2
        # reference tensors the SAC path never executes (Q/K projection rows, value head, PE buffer): kept for checkpoints
        self.passive = ckpt.default_passive(self.cfg.d_model, self.cfg.n_layers, self.cfg.head1, self.cfg.head2, seed) \
            if self.cfg.family == 0 else {}
        pe = positional_encoding_table(self.cfg.pe_rows, self.cfg.d_model) if self.cfg.family == 0 else None
        self._pe_host = pe
        self._h = C.c_void_p()
        dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        nat.check(self.L.tvc_sac_create(C.byref(self.cfg), dev_index, self.params.data_ptr(), self.grads.data_ptr(),
                                        self.adam_m.data_ptr(), self.adam_v.data_ptr(),
                                        pe.data_ptr() if pe is not None else None, C.byref(self._h)))
        if init:
            self.init_parameters(seed)

    def _critic_first(self):
        return "0.weight"

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self.L.tvc_sac_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    # -- parameters
    def view(self, name: str) -> torch.Tensor:
        off, rows, cols = self.index[name]
        return self.params[off:off + rows * cols].view(rows, cols) if cols > 1 else self.params[off:off + rows]

    def grad_view(self, name: str) -> torch.Tensor:
        off, rows, cols = self.index[name]
        return self.grads[off:off + rows * cols].view(rows, cols) if cols > 1 else self.grads[off:off + rows]

    def init_parameters(self, seed: int = 0):
        """The reference's initialisation: policy Linear layers orthogonal(gain sqrt 2) + zero bias
        (agent/...:185-190), LayerNorm (1, 0), critics torch's default nn.Linear init; targets = copies."""
        g = torch.Generator().manual_seed(seed)
        for name, off, rows, cols in self.table:
            if name.startswith("target_"):
                continue
            t = torch.empty(rows, cols) if cols > 1 else torch.empty(rows)
            is_ln = cols == 1 and name.endswith("weight")
            if is_ln:
                t.fill_(1.0)
            elif cols == 1:
                if name.startswith("policy.") or self._is_ln_bias(name):
                    t.zero_()
                else:  # nn.Linear default bias: U(-1/sqrt(fan_in), 1/sqrt(fan_in))
                    fan_in = self.index[name.replace("bias", "weight")][2]
                    t.uniform_(-1 / math.sqrt(fan_in), 1 / math.sqrt(fan_in), generator=g)
            elif name.startswith("policy."):
                torch.nn.init.orthogonal_(t, gain=math.sqrt(2), generator=g)
            else:  # kaiming_uniform(a=sqrt 5) == U(-1/sqrt(fan_in), 1/sqrt(fan_in))
                t.uniform_(-1 / math.sqrt(cols), 1 / math.sqrt(cols), generator=g)
            self.view(name).copy_(t)
        self.sync_targets()

    def _is_ln_bias(self, name):
        w = name.replace("bias", "weight")
        return w in self.index and self.index[w][2] == 1

    def sync_targets(self):
        n0, nc = self.n_policy, self.n_critic
        self.params[n0 + 2 * nc:n0 + 4 * nc].copy_(self.params[n0:n0 + 2 * nc])
        self.sync_derived()

    def adam_steps(self):
        out = (C.c_int32 * 2)()
        nat.check(self.L.tvc_sac_get_adam_steps(self._h, out))
        return [int(out[0]), int(out[1])]

    def set_adam_steps(self, steps):
        nat.check(self.L.tvc_sac_set_adam_steps(self._h, (C.c_int32 * 2)(int(steps[0]), int(steps[1]))))

    def act_counter(self) -> int:
        """call counter of the train-mode acting passes (keys their dropout masks); synchronises"""
        out = C.c_int32()
        nat.check(self.L.tvc_sac_get_act_counter(self._h, C.byref(out)))
        return int(out.value)

    def set_act_counter(self, value: int):
        nat.check(self.L.tvc_sac_set_act_counter(self._h, int(value)))

    def sync_derived(self):
        """call after writing `params` from the host side (folded acting weights are cached in the handle)"""
        nat.check(self.L.tvc_sac_sync_derived(self._h, self._stream()))

    def load_named(self, tensors: Dict[str, torch.Tensor]):
        """native names -> values"""
        for k, v in tensors.items():
            dst = self.view(k)
            dst.copy_(torch.as_tensor(v, dtype=torch.float32).reshape(dst.shape))
        self.sync_derived()

    def load_reference_state(self, net: str, state_dict: Dict[str, torch.Tensor]):
        """Import one reference net's state_dict (net in policy, q1, q2, target_q1, target_q2)."""
        d = self.cfg.d_model
        for k, v in state_dict.items():
            v = torch.as_tensor(v, dtype=torch.float32)
            if net == "policy" and (k.startswith("value_head") or k.startswith("pos_encoding")):
                self.passive[f"{net}.{k}"] = v.clone().cpu()
                continue
            name, sl = _ref_to_native_name(net, k)
            if sl == "v_rows":
                self.passive[f"{net}.{k}"] = v.clone().cpu()  # keeps the dead Q/K rows for export
                v = v[2 * d:3 * d]
            if name not in self.index:
                raise KeyError(f"{net}.{k} has no native tensor ({name})")
            self.view(name).copy_(v.reshape(self.view(name).shape))
        self.sync_derived()

    def export_reference_state(self, net: str) -> Dict[str, torch.Tensor]:
        """Reference-keyed state_dict of one net (checkpoint compatibility, agent/...:1098-1141)."""
        flat = self.params.detach().cpu()
        return {k: ckpt._expand(self.layout, flat, k, native, self.passive) for k, native, _ in self.layout.net_keys(net)}

    def export_checkpoint_entry(self) -> dict:
        """checkpoint['algorithms']['sac'] exactly as the reference's save_checkpoint lays it out (agent/...:1114-1125)"""
        torch.cuda.synchronize(self.device)
        return ckpt.pack_sac(self.layout, self.params, self.adam_m, self.adam_v, self.adam_steps(), self.passive)

    def import_checkpoint_entry(self, entry: dict):
        """the inverse; accepts files written by the reference itself (value_head / Q-K rows are kept for re-export)"""
        p, m, v = self.params.detach().cpu(), self.adam_m.detach().cpu(), self.adam_v.detach().cpu()
        steps, passive, have_opt = ckpt.unpack_sac(self.layout, entry, p, m, v)
        self.params.copy_(p)
        self.passive.update({k: t for k, t in passive.items() if k.startswith("policy.")})
        if have_opt:
            self.adam_m.copy_(m)
            self.adam_v.copy_(v)
            self.set_adam_steps(steps)
        self.sync_derived()
        return have_opt

    # -- hot path
    def snapshot_policy(self):
        """copy the policy into the acting snapshot (read by act(..., snapshot=True)) on the current stream"""
        nat.check(self.L.tvc_sac_snapshot_policy(self._h, self._stream()))

    def act(self, obs: torch.Tensor, eps: Optional[torch.Tensor] = None, out=None, clamp: bool = True, snapshot: bool = False,
            share_cus: bool = False, train_mode: bool = False, x3: bool = False):
        """-> (action[n,A] clamped to [-1,1] unless clamp=False, mean, log_std); eps None = deterministic.
        snapshot=True acts with the parameters of the last snapshot_policy() instead of the live ones; share_cus=True leaves
        half of every CU to other streams (tvc_sac_act flags bit 2); train_mode=True keeps Dropout active while acting, as the
        reference does (it never calls .eval(), agent/...:765): flags bit 3, per-layer kernels, fresh masks every call;
        x3=True runs the one-launch kernel (n >= 16 384 rows) on the bf16 matrix pipe with three-term split operands (flags bit 4:
        fp32-exact, tvc_actor_x3.h)."""
        n, A = obs.shape[0], self.cfg.act_dim
        assert obs.dtype == torch.float32 and obs.is_contiguous() and obs.shape[1] == self.cfg.obs_dim
        if out is None:
            out = tuple(torch.empty((n, A), dtype=torch.float32, device=self.device) for _ in range(3))
        act, mean, ls = out
        nat.check(self.L.tvc_sac_act(self._h, obs.data_ptr(), n, nat.ptr(eps), act.data_ptr(), mean.data_ptr(), ls.data_ptr(),
                                     (0 if clamp else 1) | (2 if snapshot else 0) | (4 if share_cus else 0) | (8 if train_mode else 0) | (16 if x3 else 0),
                                     self._stream()))
        return act, mean, ls

    def enable_x3(self):
        """Packs the split-operand weight stream now (tvc_sac_enable_x3): call before a loop whose updates run on another stream."""
        nat.check(self.L.tvc_sac_enable_x3(self._h, self._stream()))

    def update(self, s, a, r, s2, d, eps_next, eps_new, all_reduce=None, grad_scale: float = 1.0):
        """One _update_sac.  all_reduce(tensor) is called on the critic and then on the actor gradient
        slices between the grads and apply phases (data-parallel training); losses stay on the device."""
        L, h, st = self.L, self._h, self._stream()
        if all_reduce is None:
            nat.check(L.tvc_sac_update(h, s.data_ptr(), a.data_ptr(), r.data_ptr(), s2.data_ptr(), d.data_ptr(),
                                       eps_next.data_ptr(), eps_new.data_ptr(), self.losses.data_ptr(), st))
            return self.losses
        nat.check(L.tvc_sac_critic_grads(h, s.data_ptr(), a.data_ptr(), r.data_ptr(), s2.data_ptr(), d.data_ptr(),
                                         eps_next.data_ptr(), self.losses.data_ptr(), st))
        all_reduce(self.grads[self.n_policy:])
        nat.check(L.tvc_sac_critic_apply(h, grad_scale, st))
        nat.check(L.tvc_sac_actor_grads(h, s.data_ptr(), eps_new.data_ptr(), self.losses.data_ptr(), st))
        all_reduce(self.grads[:self.n_policy])
        nat.check(L.tvc_sac_actor_apply(h, grad_scale, st))
        return self.losses

    # the four phases of update(), for callers that interleave them with other work / collectives
    def critic_grads(self, s, a, r, s2, d, eps_next):
        nat.check(self.L.tvc_sac_critic_grads(self._h, s.data_ptr(), a.data_ptr(), r.data_ptr(), s2.data_ptr(), d.data_ptr(),
                                              eps_next.data_ptr(), self.losses.data_ptr(), self._stream()))

    def critic_apply(self, grad_scale: float = 1.0):
        nat.check(self.L.tvc_sac_critic_apply(self._h, grad_scale, self._stream()))

    def actor_grads(self, s, eps_new):
        nat.check(self.L.tvc_sac_actor_grads(self._h, s.data_ptr(), eps_new.data_ptr(), self.losses.data_ptr(), self._stream()))

    def actor_apply(self, grad_scale: float = 1.0):
        nat.check(self.L.tvc_sac_actor_apply(self._h, grad_scale, self._stream()))

    def q_values(self, s, a, target=False):
        q = torch.empty((2, s.shape[0]), dtype=torch.float32, device=self.device)
        nat.check(self.L.tvc_sac_q_values(self._h, s.data_ptr(), a.data_ptr(), s.shape[0], 1 if target else 0, q.data_ptr(),
                                          self._stream()))
        return q


class ReplayBuffer:
    """Device-resident uniform replay (capacity 1M x 96 B = 96 MB at the BASELINE shapes)."""

    def __init__(self, capacity: int, obs_dim: int = 10, act_dim: int = 2, device="cuda:0", seed: int = 0):
        self.L = nat.load()
        self.device = torch.device(device)
        self.capacity, self.obs_dim, self.act_dim, self.seed = capacity, obs_dim, act_dim, seed
        self._h = C.c_void_p()
        dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        nat.check(self.L.tvc_replay_create(capacity, obs_dim, act_dim, dev_index, C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self.L.tvc_replay_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self):
        return int(self.L.tvc_replay_size(self._h))

    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def insert(self, s, a, r, s2, term, trunc=None):
        n = s.shape[0]
        nat.check(self.L.tvc_replay_insert(self._h, s.data_ptr(), a.data_ptr(), r.data_ptr(), s2.data_ptr(), term.data_ptr(),
                                           nat.ptr(trunc), n, self._stream()))

    def store_transition(self, obs, action, reward, next_obs, done):
        """legacy single-transition surface (tests/test_agent.py:99-108)"""
        dev = self.device
        f = lambda x, n: torch.as_tensor(np.asarray(x, dtype=np.float32).reshape(1, n), device=dev)
        self.insert(f(obs, self.obs_dim), f(action, self.act_dim), torch.tensor([float(reward)], device=dev),
                    f(next_obs, self.obs_dim), torch.tensor([1 if done else 0], dtype=torch.uint8, device=dev))

    def export(self):
        """-> (rows[size, 2*obs+A+2] in storage order, [head, size, sample counter]); synchronises (checkpoints only)"""
        meta = (C.c_int64 * 3)()
        nat.check(self.L.tvc_replay_export(self._h, None, meta))
        rows = torch.empty((int(meta[1]), 2 * self.obs_dim + self.act_dim + 2), dtype=torch.float32, device=self.device)
        nat.check(self.L.tvc_replay_export(self._h, rows.data_ptr() if meta[1] > 0 else None, meta))
        return rows, [int(meta[0]), int(meta[1]), int(meta[2])]

    def import_(self, rows: torch.Tensor, meta):
        rows = torch.as_tensor(rows, dtype=torch.float32).to(self.device).contiguous()
        assert rows.shape[0] == int(meta[1]) and (rows.numel() == 0 or rows.shape[1] == 2 * self.obs_dim + self.act_dim + 2)
        nat.check(self.L.tvc_replay_import(self._h, rows.data_ptr() if rows.numel() else None,
                                           (C.c_int64 * 3)(int(meta[0]), int(meta[1]), int(meta[2]))))

    def sample(self, batch: int, counter: Optional[int] = None, out=None):
        dev = self.device
        if out is None:
            out = (torch.empty((batch, self.obs_dim), device=dev), torch.empty((batch, self.act_dim), device=dev),
                   torch.empty((batch,), device=dev), torch.empty((batch, self.obs_dim), device=dev),
                   torch.empty((batch,), device=dev))
        s, a, r, s2, d = out
        nat.check(self.L.tvc_replay_sample(self._h, batch, self.seed, AUTO_COUNTER if counter is None else counter,
                                           s.data_ptr(), a.data_ptr(), r.data_ptr(), s2.data_ptr(), d.data_ptr(),
                                           self._stream()))
        return out


def _physics_loss_eager(s, a, s2, weight):
    """PhysicsInformedLoss.forward (agent/...:236-285) for the eager pass-through algorithms (the HIP update computes it itself)"""
    w, w2 = s[:, 4:7], s2[:, 4:7]
    an = torch.norm(a, dim=-1, keepdim=True)
    mom = torch.mean((w2 - (w + an * 0.1)) ** 2)
    en = torch.mean((0.5 * (w2 ** 2).sum(-1) - (0.5 * (w ** 2).sum(-1) + 0.5 * (a ** 2).sum(-1) * 0.01)) ** 2)
    qn = torch.mean((torch.norm(s[:, :4], dim=-1) - 1.0) ** 2) + torch.mean((torch.norm(s2[:, :4], dim=-1) - 1.0) ** 2)
    return weight * (mom + en + qn)


class MultiAlgorithmAgent:
    """Drop-in for the reference class of the same name (agent/multi_algorithm_agent.py:419).

    Same constructor ``(obs_dim, action_dim, config)`` and methods ``select_algorithm``, ``get_action``, ``update``,
    ``update_performance``, ``save_checkpoint``, ``load_checkpoint``, ``to``; attributes ``performance_history``,
    ``algorithms``, ``algorithm_weights``, ``device``.

    * 'sac' is the MI355X-native learner (``NativeSAC``: HIP kernels end to end).
    * 'ppo' and 'td3' are an eager-PyTorch PASS-THROUGH (``passthrough.py``, not accelerated), present when
      ``algorithms.<name>.enabled`` (default true, like the reference :487-497) so that ``select_algorithm`` follows the
      reference rule (:693-709) -- which, like the reference, answers 'ppo' until another algorithm has a performance history.
      ``tvc_native.passthrough: false`` leaves them out.
    * With ``hierarchical_rl.enabled`` (true in the shipped config.yaml:103-104) ``get_action`` follows the reference (:751-754)
      and acts with the never-trained goal policy + goal-conditioned low-level policy of ``hierarchical.HierarchicalPolicy``.
    Deliberate fix: ``update`` accepts the BoolTensor ``dones`` that scripts/train.py:582 builds (the reference
    raises on it inside _update_sac / _update_td3 and skips the update).
    """

    def __init__(self, obs_dim: int, action_dim: int, config: dict, device=None, seed: int = 42):
        self.obs_dim, self.action_dim, self.config = obs_dim, action_dim, config or {}
        self.logger = logging.getLogger(__name__)
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        native = self.config.get("tvc_native", {}) or {}
        net = self.config.get("network", {}) or {}
        tr = net.get("transformer", {}) or {}
        hd = (net.get("mlp_backbone", {}) or {}).get("hidden_dims", [512, 512, 256])
        family = int(native.get("family", 0))
        self.batch_size = int(native.get("batch_size", 1))
        algs = self.config.get("algorithms", {}) or {}
        enabled = lambda a: bool((algs.get(a, {}) or {}).get("enabled", True))
        passthrough_on = bool(native.get("passthrough", True))
        self.algorithms = {}
        self.algorithm_weights = {}
        # construction order of the reference (:487-497): ppo, sac, td3 -- it decides the "first available" fallback (:757-759)
        if enabled("ppo") and passthrough_on:
            from . import passthrough
            torch.manual_seed(seed + 101)
            self.algorithms["ppo"] = passthrough.make_ppo(obs_dim, action_dim, self.config, self.device)
            self.algorithm_weights["ppo"] = 1.0
        if enabled("sac"):
            cfg = sac_cfg(family, obs_dim=obs_dim, act_dim=action_dim, d_model=int(tr.get("d_model", 256)),
                          n_layers=int(tr.get("num_layers", 4)), ff_dim=int(tr.get("dim_feedforward", 512)),
                          head1=int(hd[0]), head2=int(hd[1]), batch_size=self.batch_size,
                          max_act_rows=int(native.get("max_act_rows", 4096)), pe_rows=int(native.get("pe_rows", 1)),
                          nhead=int(tr.get("nhead", 8)),
                          # the reference never calls .eval(): its update runs with Dropout active (network.transformer.dropout
                          # for the policy, a hard-coded 0.1 in the critics, agent/...:457,596-604); one value drives both here
                          dropout_p=float(native.get("dropout", tr.get("dropout", 0.1))) if family == 0 else 0.0)
            self.sac = NativeSAC(cfg, device=self.device, seed=seed)
            # ... and neither does its get_action: the policy acts with Dropout active (agent/...:765).  Mirrored by default
            # (tvc_native.acting_dropout: false = act with the deterministic net, what VecTrainer does for throughput)
            self._act_train_mode = bool(native.get("acting_dropout", True)) and cfg.dropout_p > 0.0
            self.algorithms["sac"] = {"type": "sac", "native": self.sac}
            self.algorithm_weights["sac"] = 1.0
        else:
            self.sac = None
            self._act_train_mode = False
        if enabled("td3") and passthrough_on:
            from . import passthrough
            torch.manual_seed(seed + 103)
            self.algorithms["td3"] = passthrough.make_td3(obs_dim, action_dim, self.device)
            self.algorithm_weights["td3"] = 1.0
        self.performance_history = {alg: deque(maxlen=100) for alg in ["ppo", "sac", "td3"]}
        self.selection_strategy = (algs.get("ensemble", {}) or {}).get("selection_strategy", "dynamic")
        self._gen = torch.Generator(device=self.device).manual_seed(seed)
        # SafetyLayer (agent/...:515-520): an untrained correction net + constraint test, applied when enabled
        self.safety_layer = None
        saf = self.config.get("safety", {}) or {}
        if (saf.get("safety_layer", {}) or {}).get("enabled", False):
            from .curiosity import SafetyLayer
            cons = (saf.get("constrained_rl", {}) or {}).get("constraints", {}) or {}
            self.safety_layer = SafetyLayer(device=self.device, max_rows=int(native.get("max_act_rows", 4096)),
                                            state_dim=obs_dim, action_dim=action_dim, max_tilt=cons.get("max_tilt", 0.52),
                                            max_angular_velocity=cons.get("max_angular_velocity", 5.0), seed=seed)

        # HierarchicalAgent (agent/...:499-504): acting goes through it when enabled
        self.hierarchical_agent = None
        if (self.config.get("hierarchical_rl", {}) or {}).get("enabled", False):
            from .hierarchical import HierarchicalPolicy
            self.hierarchical_agent = HierarchicalPolicy(obs_dim, action_dim, device=self.device,
                                                         max_rows=int(native.get("max_act_rows", 4096)), seed=seed,
                                                         train_mode=bool(native.get("acting_dropout", True)))

    def to(self, device):
        if torch.device(device).type != self.device.type:
            self.logger.warning("MultiAlgorithmAgent(native) stays on %s (requested %s)", self.device, device)
        return self

    def select_algorithm(self, performance_metrics: Optional[Dict] = None):
        """the reference rule, agent/...:693-709: 'dynamic' = best mean of the last 10 performances among the algorithms that
        HAVE a history, else 'ppo'; 'voting' = 'ensemble'; otherwise best mean over the whole history, else 'ppo'"""
        previous = getattr(self, "_current_algorithm", None)
        if self.selection_strategy == "voting":
            selected = "ensemble"
        else:
            last = 10 if self.selection_strategy == "dynamic" else None
            best, best_perf = None, -float("inf")
            for name, hist in self.performance_history.items():
                if len(hist) > 0 and name in self.algorithms:
                    vals = list(hist)[-last:] if last else list(hist)
                    perf = float(np.mean(vals))
                    if perf > best_perf:
                        best, best_perf = name, perf
            # the reference answers 'ppo' whenever no algorithm has a history yet (:709); where 'ppo' was not built (pass-through off,
            # algorithms.ppo.enabled: false) the first available algorithm stands in, as get_action does (:757-759), so that
            # update(batch) under the reference's driver trains what acts
            selected = best or ("ppo" if ("ppo" in self.algorithms or not self.algorithms) else next(iter(self.algorithms)))
        if previous is not None and previous != selected:
            self.logger.info(f"Algorithm switch: {previous} -> {selected}")
        self._current_algorithm = selected
        return selected

    def _policy_action(self, name: str, state: torch.Tensor, deterministic: bool, clamp: bool):
        """(action, mean, log_std) of one algorithm's policy: HIP kernels for 'sac', eager torch for the pass-through nets"""
        agent = self.algorithms[name]
        if agent["type"] == "sac":
            eps = None if deterministic else torch.randn((state.shape[0], self.action_dim), device=self.device, generator=self._gen)
            return self.sac.act(state, eps, clamp=clamp, train_mode=self._act_train_mode)
        from . import passthrough
        with torch.no_grad():
            mean, log_std, _ = passthrough.policy_outputs(agent, state)
            act = passthrough.sample(agent, mean, log_std, deterministic)
        return (torch.clamp(act, -1.0, 1.0) if clamp else act), mean, log_std

    def get_action(self, state: torch.Tensor, deterministic: bool = False, algorithm: Optional[str] = None):
        """-> (np.ndarray[B, A], info) like agent/...:736-809."""
        try:
            if algorithm is None:
                algorithm = self.select_algorithm()
            state = torch.as_tensor(state, dtype=torch.float32, device=self.device)
            if state.dim() == 1:
                state = state.unsqueeze(0)
            state = state.contiguous()
            raw = self.safety_layer is not None  # the safety layer sees the unclamped sample (agent/...:785-789)
            if algorithm == "ensemble":  # agent/...:811-866: weighted mean of every algorithm's action
                acts, weights = [], []
                for name in self.algorithms:
                    a, _, _ = self._policy_action(name, state, deterministic, clamp=False)
                    acts.append(a)
                    weights.append(self.algorithm_weights.get(name, 1.0))
                w = torch.tensor(weights, device=self.device)
                w = w / w.sum()
                act = sum(wi * ai for wi, ai in zip(w, acts))
                act = self.safety_layer.apply(state, act.contiguous()) if raw else torch.clamp(act, -1.0, 1.0)
                return act.cpu().numpy(), {"algorithm": "ensemble", "weights": w.cpu().numpy(),
                                           "individual_actions": [a.cpu().numpy() for a in acts]}
            goal = None
            if self.hierarchical_agent is not None:  # agent/...:751-754
                eps = None if deterministic else torch.randn((state.shape[0], self.action_dim), device=self.device, generator=self._gen)
                act, mean, ls, goal = self.hierarchical_agent.act(state, eps, clamp=not raw)
            else:
                if algorithm not in self.algorithms:
                    self.logger.warning(f"Algorithm {algorithm} not available, falling back to first available")
                    algorithm = list(self.algorithms.keys())[0]
                act, mean, ls = self._policy_action(algorithm, state, deterministic, clamp=not raw)
            if raw:
                act = self.safety_layer.apply(state, act.contiguous())
            info = {"algorithm": algorithm, "mean": mean.cpu().numpy(), "log_std": ls.cpu().numpy(), "value": None}
            if goal is not None:
                info["goal"] = goal.cpu().numpy()
            return act.cpu().numpy(), info
        except Exception as e:  # reference convention: log and fall back to a random action (agent/...:804-809)
            self.logger.error(f"Error in get_action: {e}")
            return np.random.uniform(-1.0, 1.0, size=(self.action_dim,)), {"algorithm": "fallback", "error": str(e)}

    def update(self, batch: Dict, algorithm: Optional[str] = None):
        """-> dict of float losses like agent/...:868-912; never raises (returns {'error': msg})."""
        try:
            if algorithm is None:
                algorithm = self.select_algorithm()
            f = lambda k, shape: torch.as_tensor(batch[k]).to(device=self.device, dtype=torch.float32).reshape(shape).contiguous()
            B = int(torch.as_tensor(batch["states"]).reshape(-1, self.obs_dim).shape[0])
            s, s2 = f("states", (B, self.obs_dim)), f("next_states", (B, self.obs_dim))
            a, r, d = f("actions", (B, self.action_dim)), f("rewards", (B,)), f("dones", (B,))
            physics = (self.config.get("physics_informed", {}) or {}).get("enabled", False)
            if algorithm == "sac" and "sac" in self.algorithms:
                if B != self.batch_size:
                    raise ValueError(f"batch of {B} rows, agent built for tvc_native.batch_size={self.batch_size}")
                e1 = torch.randn((B, self.action_dim), device=self.device, generator=self._gen)
                e2 = torch.randn((B, self.action_dim), device=self.device, generator=self._gen)
                losses = self.sac.update(s, a, r, s2, d, e1, e2).cpu().tolist()
                out = {"q1_loss": losses[0], "q2_loss": losses[1], "policy_loss": losses[2]}
                if physics and losses[3] > 0:
                    out["physics_loss"] = losses[3]
                return out
            if algorithm in ("ppo", "td3") and algorithm in self.algorithms:
                from . import passthrough
                tb = {"states": s, "actions": a, "rewards": r, "next_states": s2, "dones": d}
                out = passthrough.update_ppo(self.algorithms["ppo"], tb) if algorithm == "ppo" \
                    else passthrough.update_td3(self.algorithms["td3"], tb)
                if physics:  # PhysicsInformedLoss (:236-285), reported only
                    w = float((self.config.get("physics_informed", {}) or {}).get("physics_loss_weight", 0.1))
                    pl = float(_physics_loss_eager(s, a, s2, w))
                    if pl > 0:
                        out["physics_loss"] = pl
                return out
            self.logger.warning(f"Algorithm {algorithm} not available for update")
            return {}
        except Exception as e:
            self.logger.error(f"Error in agent update: {e}")
            return {"error": str(e)}

    def update_performance(self, algorithm: str, performance: float):
        if algorithm in self.performance_history:
            self.performance_history[algorithm].append(performance)
            if len(self.performance_history[algorithm]) >= 10:
                recent = np.mean(list(self.performance_history[algorithm])[-10:])
                self.algorithm_weights[algorithm] = max(0.1, recent)

    # checkpoint layout of agent/...:1098-1141: keys algorithms.sac.{policy_state, q1_state, q2_state, target_q1_state,
    # target_q2_state, optimizer_policy_state, optimizer_q1_state, optimizer_q2_state, type}, performance_history (dict of
    # deques, :1102), algorithm_weights, config.  The reference's own load_checkpoint reads a file written here (tested in the
    # build container against the reference class, tests/test_checkpoint_cpu.py), and a file the reference wrote loads here.
    def save_checkpoint(self, path: str):
        entries = {}
        for name, agent in self.algorithms.items():
            if agent["type"] == "sac":
                entries[name] = self.sac.export_checkpoint_entry()
            else:
                from . import passthrough
                entries[name] = passthrough.checkpoint_entry(agent)
        ckpt_dict = {
            "algorithms": entries,
            "performance_history": dict(self.performance_history),
            "algorithm_weights": self.algorithm_weights,
            "config": self.config,
        }
        torch.save(ckpt_dict, path)
        self.logger.info(f"Checkpoint saved to {path}")

    def load_checkpoint(self, path: str):
        c = ckpt.load_file(path)  # weights-only loader, deque allow-listed
        self.performance_history = {k: deque(v, maxlen=100) for k, v in c["performance_history"].items()}
        self.algorithm_weights = c["algorithm_weights"]
        for name in ("ppo", "td3"):
            e = c["algorithms"].get(name)
            if name in self.algorithms and e is not None and "policy_state" in e:
                from . import passthrough
                passthrough.load_checkpoint_entry(self.algorithms[name], e)
        entry = c["algorithms"].get("sac")
        if entry is None or self.sac is None:
            return
        have_opt = self.sac.import_checkpoint_entry(entry)
        if not have_opt and "native_adam" in entry:  # round-1 files
            self.sac.adam_m.copy_(entry["native_adam"]["m"])
            self.sac.adam_v.copy_(entry["native_adam"]["v"])
            if "steps" in entry["native_adam"]:
                self.sac.set_adam_steps(entry["native_adam"]["steps"])
        self.logger.info(f"Checkpoint loaded from {path}")
