"""Eager-PyTorch pass-through for the algorithms the hot path does NOT cover: PPO and TD3.

NOT accelerated and not part of the measured path: plain torch.nn modules and torch.optim.Adam on the agent's device, so that
``MultiAlgorithmAgent.select_algorithm()`` (agent/multi_algorithm_agent.py:693-709), ``get_action`` (:736-809, incl. the
'ensemble' strategy :811-866) and ``update`` (:868-912) keep working for every algorithm name the reference knows, and so that
checkpoints carry the reference's 'ppo' / 'td3' entries (:1108-1140).  The SAC entry is the HIP learner (agent.NativeSAC).

Behaviour mirrored, with its quirks:
  * PPO (:573-585, 914-948): one TransformerPolicyNetwork-shaped net with a value head, Adam(lr = algorithms.ppo.learning_rate,
    default 2.5e-4); the "simplified" update: advantages = rewards, loss = -(log_prob * rewards).mean() + 0.5 * mse(value, rewards),
    no clipping, no GAE, std = exp(log_std) unclamped.
  * TD3 (:629-681, 1018-1086): tanh-headed 512/256 GELU+LayerNorm+Dropout policy, twin critics, targets, three Adam(3e-4);
    target smoothing noise 0.2 clipped to 0.5, policy + Polyak (tau 0.005) every second call, policy loss through q1 only.
  * the policy net indexes its positional-encoding table by BATCH ROW, and the nets stay in train mode (Dropout
    active) while acting, exactly like the reference, which never calls .eval().
The module and parameter names equal the reference's, so its state_dicts load here and vice versa.
"""
import copy
import math
from typing import Dict, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.distributions import Normal


class PositionalTable(nn.Module):
    """buffer `pe` [max_len, 1, d] of agent/...:90-105; added by batch row, as the reference does (F9)"""

    def __init__(self, d_model: int, max_len: int = 5000):
        super().__init__()
        pos = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
        div = torch.exp(torch.arange(0, d_model, 2).float() * (-math.log(10000.0) / d_model))
        pe = torch.zeros(max_len, d_model)
        pe[:, 0::2] = torch.sin(pos * div)
        pe[:, 1::2] = torch.cos(pos * div)
        self.register_buffer("pe", pe.unsqueeze(0).transpose(0, 1))

    def forward(self, x):
        return x + self.pe[:x.size(0), :]


def _head(d_in: int, h1: int, h2: int, d_out: int, p: float) -> nn.Sequential:
    return nn.Sequential(nn.Linear(d_in, h1), nn.GELU(), nn.LayerNorm(h1), nn.Dropout(p),
                         nn.Linear(h1, h2), nn.GELU(), nn.LayerNorm(h2), nn.Dropout(p), nn.Linear(h2, d_out))


class EagerTransformerPolicy(nn.Module):
    """Same modules, names and forward as the reference's TransformerPolicyNetwork (:123-227) with the shipped NetworkConfig
    (SqueezeExcitation off): -> (mean, log_std clamped to [-20, 2], value)."""

    def __init__(self, obs_dim: int, action_dim: int, d_model: int = 256, nhead: int = 8, num_layers: int = 4, dim_ff: int = 512,
                 hidden=(512, 512), dropout: float = 0.1):
        super().__init__()
        self.input_embedding = nn.Linear(obs_dim, d_model)
        self.pos_encoding = PositionalTable(d_model)
        layer = nn.TransformerEncoderLayer(d_model=d_model, nhead=nhead, dim_feedforward=dim_ff, dropout=dropout, activation="gelu",
                                           batch_first=True)
        self.transformer_encoder = nn.TransformerEncoder(layer, num_layers=num_layers)
        self.feature_norm = nn.LayerNorm(d_model)
        self.policy_head = _head(d_model, hidden[0], hidden[1], 2 * action_dim, dropout)
        self.value_head = _head(d_model, hidden[0], hidden[1], 1, dropout)
        for m in self.modules():  # orthogonal(gain sqrt 2) + zero bias on every nn.Linear, LayerNorm (1, 0)  (:185-190)
            if isinstance(m, nn.Linear):
                nn.init.orthogonal_(m.weight, gain=math.sqrt(2))
                nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.LayerNorm):
                nn.init.constant_(m.bias, 0)
                nn.init.constant_(m.weight, 1.0)

    def forward(self, state):
        x = self.input_embedding(state)
        x = self.pos_encoding(x.unsqueeze(1))
        x = self.transformer_encoder(x)[:, -1, :]
        x = self.feature_norm(x)
        mean, log_std = torch.chunk(self.policy_head(x), 2, dim=-1)
        return mean, torch.clamp(log_std, -20, 2), self.value_head(x).squeeze(-1)


def _td3_mlp(d_in: int, d_out: int, tanh: bool) -> nn.Sequential:
    layers = [nn.Linear(d_in, 512), nn.GELU(), nn.LayerNorm(512), nn.Dropout(0.1), nn.Linear(512, 256), nn.GELU(), nn.LayerNorm(256),
              nn.Dropout(0.1), nn.Linear(256, d_out)]
    if tanh:
        layers.append(nn.Tanh())
    return nn.Sequential(*layers)


def make_ppo(obs_dim: int, action_dim: int, config: dict, device) -> dict:
    tr = (config.get("network", {}) or {}).get("transformer", {}) or {}
    hd = ((config.get("network", {}) or {}).get("mlp_backbone", {}) or {}).get("hidden_dims", [512, 512, 256])
    net = EagerTransformerPolicy(obs_dim, action_dim, int(tr.get("d_model", 256)), int(tr.get("nhead", 8)), int(tr.get("num_layers", 4)),
                                 int(tr.get("dim_feedforward", 512)), (int(hd[0]), int(hd[1])), float(tr.get("dropout", 0.1))).to(device)
    lr = float(((config.get("algorithms", {}) or {}).get("ppo", {}) or {}).get("learning_rate", 2.5e-4))
    return {"policy": net, "optimizer": torch.optim.Adam(net.parameters(), lr=lr), "type": "ppo", "eager": True}


def make_td3(obs_dim: int, action_dim: int, device) -> dict:
    pol = _td3_mlp(obs_dim, action_dim, True).to(device)
    q1 = _td3_mlp(obs_dim + action_dim, 1, False).to(device)
    q2 = _td3_mlp(obs_dim + action_dim, 1, False).to(device)
    return {"policy": pol, "q1": q1, "q2": q2, "target_policy": copy.deepcopy(pol), "target_q1": copy.deepcopy(q1),
            "target_q2": copy.deepcopy(q2), "optimizer_policy": torch.optim.Adam(pol.parameters(), lr=3e-4),
            "optimizer_q1": torch.optim.Adam(q1.parameters(), lr=3e-4), "optimizer_q2": torch.optim.Adam(q2.parameters(), lr=3e-4),
            "type": "td3", "eager": True, "update_counter": 0}


def policy_outputs(agent: dict, state: torch.Tensor):
    """-> (mean, log_std, value) the way get_action reads each algorithm type (:765-769)"""
    if agent["type"] == "td3":
        mean = agent["policy"](state)
        return mean, torch.zeros_like(mean), None
    return agent["policy"](state)


def sample(agent: dict, mean, log_std, deterministic: bool):
    if deterministic:
        return mean
    if agent["type"] == "td3":
        return mean + torch.randn_like(mean) * 0.1
    return Normal(mean, torch.exp(torch.clamp(log_std, -20, 2))).sample()


def update_ppo(agent: dict, batch: Dict[str, torch.Tensor]) -> Dict[str, float]:
    states, actions, rewards = batch["states"], batch["actions"], batch["rewards"]
    mean, log_std, values = agent["policy"](states)
    log_probs = Normal(mean, torch.exp(log_std)).log_prob(actions).sum(dim=-1)
    policy_loss = -(log_probs * rewards).mean()
    value_loss = F.mse_loss(values, rewards)
    total = policy_loss + 0.5 * value_loss
    agent["optimizer"].zero_grad()
    total.backward()
    agent["optimizer"].step()
    return {"policy_loss": policy_loss.item(), "value_loss": value_loss.item(), "total_loss": total.item()}


def update_td3(agent: dict, batch: Dict[str, torch.Tensor]) -> Dict[str, float]:
    s, a, r, s2, d = batch["states"], batch["actions"], batch["rewards"], batch["next_states"], batch["dones"]
    with torch.no_grad():
        noise = torch.clamp(torch.randn_like(a) * 0.2, -0.5, 0.5)
        a2 = torch.clamp(agent["target_policy"](s2) + noise, -1.0, 1.0)
        x2 = torch.cat([s2, a2], dim=-1)
        y = r + 0.99 * (1 - d) * torch.min(agent["target_q1"](x2), agent["target_q2"](x2)).squeeze()
    x = torch.cat([s, a], dim=-1)
    q1_loss = F.mse_loss(agent["q1"](x).squeeze(), y)
    q2_loss = F.mse_loss(agent["q2"](x).squeeze(), y)
    for opt, loss in (("optimizer_q1", q1_loss), ("optimizer_q2", q2_loss)):
        agent[opt].zero_grad()
        loss.backward()
        agent[opt].step()
    policy_loss: Optional[torch.Tensor] = None
    agent["update_counter"] += 1
    if agent["update_counter"] % 2 == 0:
        policy_loss = -agent["q1"](torch.cat([s, agent["policy"](s)], dim=-1)).mean()
        agent["optimizer_policy"].zero_grad()
        policy_loss.backward()
        agent["optimizer_policy"].step()
        with torch.no_grad():
            for tgt, src in (("target_policy", "policy"), ("target_q1", "q1"), ("target_q2", "q2")):
                for tp, p in zip(agent[tgt].parameters(), agent[src].parameters()):
                    tp.mul_(1 - 0.005).add_(p, alpha=0.005)
    return {"q1_loss": q1_loss.item(), "q2_loss": q2_loss.item(), "policy_loss": policy_loss.item() if policy_loss is not None else 0.0}


def checkpoint_entry(agent: dict) -> dict:
    """the reference's per-algorithm checkpoint dict (:1108-1139)"""
    if agent["type"] == "ppo":
        return {"policy_state": agent["policy"].state_dict(), "optimizer_state": agent["optimizer"].state_dict(), "type": "ppo"}
    out = {f"{k}_state": agent[k].state_dict() for k in ("policy", "q1", "q2", "target_policy", "target_q1", "target_q2")}
    out.update({f"optimizer_{k}_state": agent[f"optimizer_{k}"].state_dict() for k in ("policy", "q1", "q2")})
    out["type"] = "td3"
    return out


def load_checkpoint_entry(agent: dict, entry: dict):
    if agent["type"] == "ppo":
        agent["policy"].load_state_dict(entry["policy_state"])
        agent["optimizer"].load_state_dict(entry["optimizer_state"])
        return
    for k in ("policy", "q1", "q2", "target_policy", "target_q1", "target_q2"):
        agent[k].load_state_dict(entry[f"{k}_state"])
    for k in ("policy", "q1", "q2"):
        agent[f"optimizer_{k}"].load_state_dict(entry[f"optimizer_{k}_state"])
