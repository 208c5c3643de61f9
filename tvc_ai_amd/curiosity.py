"""Acting-path extras of the reference as device kernels (csrc/tvc_sac.hip, tvc_mlp_*):

* ``SmallMLP``      -- dims[0] -> ... -> dims[-1] MLP handle (ReLU / GELU, optional LayerNorms) with torch-owned parameters.
* ``VecCuriosity``  -- CuriosityModule.compute_intrinsic_reward (env/enhanced_rocket_tvc_env.py:226-269) for N envs:
                       the reference builds the forward model with torch's default init and never trains it.
* ``SafetyLayer``   -- SafetyLayer.forward (agent/multi_algorithm_agent.py:287-351) + get_action's clamp.
"""
import ctypes as C
import math
from typing import Dict, Optional, Sequence

import torch

from . import _native as nat


MLP_LAYERNORM = 0x100  # TVC_MLP_LAYERNORM of include/tvc_native.h


class SmallMLP:
    """dims[0] -> ... -> dims[-1]; act 1 GELU / 2 ReLU between layers; layernorm=True puts a LayerNorm behind every hidden
    activation (nn.Sequential numbering Linear, act, LayerNorm, Linear, ...)."""

    def __init__(self, dims: Sequence[int], device="cuda:0", max_rows: int = 4096, act: int = 2, seed: int = 0,
                 layernorm: bool = False):
        self.L = nat.load()
        self.dims = [int(d) for d in dims]
        self.n_layers = len(self.dims) - 1
        self.layernorm = bool(layernorm)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise nat.TvcError("SmallMLP needs a GPU device: there is no CPU fallback")
        self._dims_c = (C.c_int32 * len(self.dims))(*self.dims)
        flags = act | (MLP_LAYERNORM if layernorm else 0)
        n = self.L.tvc_mlp_layout(self._dims_c, self.n_layers, flags, -1, None, None, None, None)
        if n < 0:
            raise nat.TvcError(self.L.tvc_last_error().decode())
        self.params = torch.zeros(n, dtype=torch.float32, device=self.device)
        self.offsets = []
        for l in range(self.n_layers):
            w, b, gw, gb = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
            if self.L.tvc_mlp_layout(self._dims_c, self.n_layers, flags, l, C.byref(w), C.byref(b), C.byref(gw), C.byref(gb)) < 0:
                raise nat.TvcError(self.L.tvc_last_error().decode())
            self.offsets.append((w.value, b.value, gw.value, gb.value))
        self.max_rows = max_rows
        self._h = C.c_void_p()
        dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        nat.check(self.L.tvc_mlp_create(self._dims_c, self.n_layers, flags, max_rows, dev_index, self.params.data_ptr(), C.byref(self._h)))
        self.init_default(seed)

    def weight(self, l):
        o, i = self.dims[l + 1], self.dims[l]
        return self.params[self.offsets[l][0]:self.offsets[l][0] + o * i].view(o, i)

    def bias(self, l):
        return self.params[self.offsets[l][1]:self.offsets[l][1] + self.dims[l + 1]]

    def ln_weight(self, l):
        return self.params[self.offsets[l][2]:self.offsets[l][2] + self.dims[l + 1]]

    def ln_bias(self, l):
        return self.params[self.offsets[l][3]:self.offsets[l][3] + self.dims[l + 1]]

    def init_default(self, seed=0):
        """torch's default nn.Linear init: U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for weight and bias; LayerNorm (1, 0)"""
        g = torch.Generator().manual_seed(seed)
        for l in range(self.n_layers):
            k = 1.0 / math.sqrt(self.dims[l])
            self.weight(l).copy_(torch.empty(self.dims[l + 1], self.dims[l]).uniform_(-k, k, generator=g))
            self.bias(l).copy_(torch.empty(self.dims[l + 1]).uniform_(-k, k, generator=g))
            if self.layernorm and l < self.n_layers - 1:
                self.ln_weight(l).fill_(1.0)
                self.ln_bias(l).zero_()

    def load_state_dict(self, sd: Dict[str, torch.Tensor]):
        """nn.Sequential numbering of the reference: '0.weight', '0.bias', '2.weight', ... (with LayerNorms: 0, 2 (LN), 3, 5 (LN), 6)"""
        per = 3 if self.layernorm else 2
        for l in range(self.n_layers):
            self.weight(l).copy_(torch.as_tensor(sd[f"{per * l}.weight"], dtype=torch.float32))
            self.bias(l).copy_(torch.as_tensor(sd[f"{per * l}.bias"], dtype=torch.float32))
            if self.layernorm and l < self.n_layers - 1:
                self.ln_weight(l).copy_(torch.as_tensor(sd[f"{per * l + 2}.weight"], dtype=torch.float32))
                self.ln_bias(l).copy_(torch.as_tensor(sd[f"{per * l + 2}.bias"], dtype=torch.float32))

    def state_dict(self) -> Dict[str, torch.Tensor]:
        per = 3 if self.layernorm else 2
        out = {}
        for l in range(self.n_layers):
            out[f"{per * l}.weight"], out[f"{per * l}.bias"] = self.weight(l).cpu().clone(), self.bias(l).cpu().clone()
            if self.layernorm and l < self.n_layers - 1:
                out[f"{per * l + 2}.weight"], out[f"{per * l + 2}.bias"] = self.ln_weight(l).cpu().clone(), self.ln_bias(l).cpu().clone()
        return out

    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def forward(self, x: torch.Tensor, x2: Optional[torch.Tensor] = None):
        n = x.shape[0]
        out = torch.empty((n, self.dims[-1]), dtype=torch.float32, device=self.device)
        k1 = self.dims[0] if x2 is None else self.dims[0] - x2.shape[1]
        nat.check(self.L.tvc_mlp_forward(self._h, x.data_ptr(), x.stride(0), k1, nat.ptr(x2), x2.stride(0) if x2 is not None else 0,
                                         n, out.data_ptr(), self._stream()))
        return out

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self.L.tvc_mlp_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class VecCuriosity(SmallMLP):
    """forward model Linear(10,256)-ReLU-Linear(256,256)-ReLU-Linear(256,8), env/...:243-249"""

    def __init__(self, device="cuda:0", max_rows: int = 4096, obs_dim: int = 8, action_dim: int = 2, hidden: int = 256, seed: int = 0):
        super().__init__([obs_dim + action_dim, hidden, hidden, obs_dim], device=device, max_rows=max_rows, act=2, seed=seed)
        self.obs_dim, self.action_dim = obs_dim, action_dim

    def add_intrinsic_reward(self, prev_obs, action, obs, rew, skip=None):
        """rew[m] += 0.01 * mse(f([prev_obs[m,:8] | action[m]]), obs[m,:8]) unless skip[m]; in place, on the stream."""
        n = rew.shape[0]
        nat.check(self.L.tvc_curiosity_add(self._h, prev_obs.data_ptr(), prev_obs.stride(0), action.data_ptr(), self.action_dim,
                                           obs.data_ptr(), nat.ptr(skip), rew.data_ptr(), n, self._stream()))
        return rew

    def intrinsic_reward(self, prev_obs, action, obs):
        rew = torch.zeros(prev_obs.shape[0], dtype=torch.float32, device=self.device)
        if obs.stride(0) != prev_obs.stride(0):
            obs = obs.contiguous()
            prev_obs = prev_obs.contiguous()
        return self.add_intrinsic_reward(prev_obs, action.contiguous(), obs, rew)


class SafetyLayer(SmallMLP):
    """safety_net Linear(12,128)-ReLU-Linear(128,64)-ReLU-Linear(64,2), agent/...:296-302; constraints :81-88"""

    def __init__(self, device="cuda:0", max_rows: int = 4096, state_dim: int = 10, action_dim: int = 2, max_tilt: float = 0.52,
                 max_angular_velocity: float = 5.0, max_control_effort: float = 1.0, seed: int = 0):
        super().__init__([state_dim + action_dim, 128, 64, action_dim], device=device, max_rows=max_rows, act=2, seed=seed)
        self.state_dim, self.action_dim = state_dim, action_dim
        self.limits = (float(max_tilt), float(max_angular_velocity), float(max_control_effort))

    def apply(self, state: torch.Tensor, proposed: torch.Tensor, out: Optional[torch.Tensor] = None):
        n = state.shape[0]
        if out is None:
            out = torch.empty((n, self.action_dim), dtype=torch.float32, device=self.device)
        nat.check(self.L.tvc_safety_apply(self._h, state.data_ptr(), self.state_dim, proposed.data_ptr(), out.data_ptr(), n,
                                          *self.limits, self._stream()))
        return out
