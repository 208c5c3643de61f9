"""Eager-PyTorch fp32 restatement of the reference SAC learner (test infrastructure only).

It is the checker for the HIP learner kernels: pinned against golden vectors taken from the reference's
own code (tests/golden/sac_ref.npz, tests/test_sac_oracle_golden.py), then used on the GPU box -- where
/root/reference does not exist -- as the fp32 reference the kernels are compared with.

Follows, with parameters addressed by the reference's state_dict names:
  TransformerPolicyNetwork.forward   agent/multi_algorithm_agent.py:192-227 (seq-len 1: attention == out_proj(v_proj(x)),
                                     SURVEY F8; positional encoding row-indexed by BATCH position)
  critics                            :593-615
  _update_sac                        :950-1016   (gamma .99, alpha .2, tau .005, three Adam(lr 3e-4))
  PhysicsInformedLoss.forward        :236-285
Never imported by the product package.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

GAMMA, ALPHA, TAU, LR = 0.99, 0.2, 0.005, 3e-4


def positional_encoding(n_rows, d_model=256):
    """PositionalEncoding buffer rows 0..n_rows-1 (agent/...:93-102)."""
    pe = torch.zeros(n_rows, d_model)
    position = torch.arange(0, n_rows, dtype=torch.float).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d_model, 2).float() * (-np.log(10000.0) / d_model))
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe


def _ln(x, P, name, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1],), P[name + ".weight"], P[name + ".bias"], eps)


class DropMasks:
    """The counter-hash dropout masks of the HIP kernels (tvc_nn_kernels.h: drop_mix / drop_key / drop_factor) restated with
    numpy, so that a train-mode update can be compared mask for mask.  p = thresh / 65536; kept values scale by 1 / (1 - p)."""

    def __init__(self, p, seed=0):
        self.seed = int(seed) & 0xFFFFFFFF  # tvc_sac_cfg.dropout_seed of the handle under test
        self.thresh = int(round(p * 65536.0))
        self.scale = np.float32(65536.0) / np.float32(65536 - self.thresh)

    @staticmethod
    def _mix(x):
        m = np.uint64(0xFFFFFFFF)
        x = x & m
        x ^= x >> np.uint64(16)
        x = (x * np.uint64(0x85EBCA6B)) & m
        x ^= x >> np.uint64(13)
        x = (x * np.uint64(0xC2B2AE35)) & m
        x ^= x >> np.uint64(16)
        return x

    def factor(self, ctr, site, z, row0, rows, cols, group=1):
        m = np.uint64(0xFFFFFFFF)
        key = self._mix(np.uint64(ctr) ^ ((np.uint64(site) * np.uint64(0x9E3779B9)) & m) ^ ((np.uint64(z) * np.uint64(0x7F4A7C15)) & m)
                        ^ np.uint64(self.seed))
        r = self._mix((np.arange(row0, row0 + rows, dtype=np.uint64) + np.uint64(0x632BE5AB)) & m)[:, None]
        c = (np.arange(cols, dtype=np.uint64) // np.uint64(group))[None, :]
        x = self._mix(key ^ r ^ (((c >> np.uint64(1)) * np.uint64(0x9E3779B1)) & m))
        b = np.where((c & np.uint64(1)) == 1, x >> np.uint64(16), x & np.uint64(0xFFFF))
        return torch.from_numpy(np.where(b >= self.thresh, self.scale, np.float32(0.0)).astype(np.float32))

    def hook(self, ctr, site_base, z=0, row0=0):
        """-> drop(op_index, x, group): x * mask for the forward call identified by (ctr, site_base, z, row0)"""
        return lambda op, x, group=1: x * self.factor(ctr, site_base + op, z, row0, x.shape[0], x.shape[1], group).to(x)


def actor_forward(P, obs, batch_pe=False, n_layers=4, drop=None, nhead=8):
    """-> mean[B,A], log_std[B,A] (clamped to [-20, 2]).  batch_pe=True reproduces the reference's
    row-indexed positional encoding (row b gets PE(b)); False = PE(0) on every row (== reference at B=1).
    drop(op_index, x, group): train-mode dropout hook (DropMasks.hook); op indices follow the native net."""
    if drop is None:
        drop = lambda op, x, group=1: x
    d = P["input_embedding.weight"].shape[0]
    x = obs @ P["input_embedding.weight"].T + P["input_embedding.bias"]
    pe = positional_encoding(obs.shape[0] if batch_pe else 1, d).to(obs)
    x = x + pe
    for l in range(n_layers):
        pre = f"transformer_encoder.layers.{l}."
        Wv = P[pre + "self_attn.in_proj_weight"][2 * d:3 * d]
        bv = P[pre + "self_attn.in_proj_bias"][2 * d:3 * d]
        v = drop(1 + 6 * l, x @ Wv.T + bv, d // nhead)  # attention-weight dropout at one key: a whole head of V
        a = drop(2 + 6 * l, v @ P[pre + "self_attn.out_proj.weight"].T + P[pre + "self_attn.out_proj.bias"])
        x = _ln(x + a, P, pre + "norm1")
        f = drop(4 + 6 * l, F.gelu(x @ P[pre + "linear1.weight"].T + P[pre + "linear1.bias"]))
        f = drop(5 + 6 * l, f @ P[pre + "linear2.weight"].T + P[pre + "linear2.bias"])
        x = _ln(x + f, P, pre + "norm2")
    base = 1 + 6 * n_layers + (2 if "se_block.fc1.weight" in P else 0)  # the SE block is two ops of the native net
    x = _ln(x, P, "feature_norm")
    if "se_block.fc1.weight" in P:  # SqueezeExcitation (agent/...:104-118): pooling a [B, C, 1] tensor over its last axis is the identity
        y = F.relu(x @ P["se_block.fc1.weight"].T + P["se_block.fc1.bias"])
        x = x * torch.sigmoid(y @ P["se_block.fc2.weight"].T + P["se_block.fc2.bias"])
    h = drop(base + 2, _ln(F.gelu(x @ P["policy_head.0.weight"].T + P["policy_head.0.bias"]), P, "policy_head.2"))
    h = drop(base + 4, _ln(F.gelu(h @ P["policy_head.4.weight"].T + P["policy_head.4.bias"]), P, "policy_head.6"))
    out = h @ P["policy_head.8.weight"].T + P["policy_head.8.bias"]
    a_dim = out.shape[1] // 2
    return out[:, :a_dim], torch.clamp(out[:, a_dim:], -20, 2)


def critic_forward(Q, s, a, drop=None):
    if drop is None:
        drop = lambda op, x, group=1: x
    x = torch.cat([s, a], -1)
    h = drop(1, _ln(F.gelu(x @ Q["0.weight"].T + Q["0.bias"]), Q, "2"))
    h = drop(3, _ln(F.gelu(h @ Q["4.weight"].T + Q["4.bias"]), Q, "6"))
    return (h @ Q["8.weight"].T + Q["8.bias"]).squeeze(-1)


def physics_loss(states, actions, next_states, weight=0.1):
    """agent/...:236-285; returns (total, [momentum, energy, quat_norm])."""
    w, w2 = states[:, 4:7], next_states[:, 4:7]
    ctrl = actions.norm(dim=1, keepdim=True).repeat(1, 3) * 0.1
    mom = F.mse_loss(w2, w + ctrl)
    ke, ke2 = 0.5 * (w ** 2).sum(1), 0.5 * (w2 ** 2).sum(1)
    en = F.mse_loss(ke2, ke + 0.5 * (actions ** 2).sum(1) * 0.01)
    ones = torch.ones(states.shape[0])
    qn = F.mse_loss(states[:, :4].norm(dim=1), ones) + F.mse_loss(next_states[:, :4].norm(dim=1), ones)
    return (mom + en + qn) * weight, [mom, en, qn]


class AdamState:
    """torch.optim.Adam defaults (betas .9/.999, eps 1e-8, no weight decay), written out."""

    def __init__(self, params):
        self.m = {k: torch.zeros_like(v) for k, v in params.items()}
        self.v = {k: torch.zeros_like(v) for k, v in params.items()}
        self.t = {k: 0 for k in params}

    def step(self, params, grads, lr=LR, b1=0.9, b2=0.999, eps=1e-8):
        with torch.no_grad():
            for k, g in grads.items():
                if g is None:  # parameters outside the loss graph (value head) are skipped by torch too
                    continue
                self.t[k] += 1
                t = self.t[k]
                self.m[k].mul_(b1).add_(g, alpha=1 - b1)
                self.v[k].mul_(b2).addcmul_(g, g, value=1 - b2)
                bc1, bc2 = 1 - b1 ** t, 1 - b2 ** t
                denom = (self.v[k].sqrt() / math.sqrt(bc2)).add_(eps)
                params[k].addcdiv_(self.m[k], denom, value=-lr / bc1)


class SacOracle:
    """State of one SAC learner in reference parameterisation."""

    def __init__(self, policy, q1, q2, batch_pe=False, actor_fn=None, critic_fn=None, dropout_p=0.0, dropout_seed=0):
        """dropout_p > 0: the reference's train-mode update with the HIP kernels' masks (DropMasks); site bases as in
        tvc_sac.hip: actor 0 (rows of s' offset by the batch size: one stacked forward), critics 120 / 140 / 160."""
        self.masks = DropMasks(dropout_p, dropout_seed) if dropout_p > 0 else None
        self.updates = 0
        if self.masks is None:
            self.actor_fn = actor_fn or (lambda P, x, row0=0: actor_forward(P, x, batch_pe))
            self.critic_fn = critic_fn or (lambda Q, s, a, call=0, z=0: critic_forward(Q, s, a))
        else:
            self.actor_fn = lambda P, x, row0=0: actor_forward(P, x, batch_pe, drop=self.masks.hook(self.updates, 0, 0, row0))
            self.critic_fn = lambda Q, s, a, call=0, z=0: critic_forward(Q, s, a, drop=self.masks.hook(self.updates, 100 + 20 * call, z))
        self.P = {k: v.clone().requires_grad_(True) for k, v in policy.items()}
        self.Q = [{k: v.clone().requires_grad_(True) for k, v in q.items()} for q in (q1, q2)]
        self.TQ = [{k: v.clone() for k, v in q.items()} for q in (q1, q2)]
        self.opt_p = AdamState(self.P)
        self.opt_q = [AdamState(q) for q in self.Q]
        self.batch_pe = batch_pe

    def update(self, s, a, r, s2, d, eps_next, eps_new):
        """One _update_sac (agent/...:950-1016) with the two Gaussian draws supplied."""
        fancy = self.masks is not None
        kw = (lambda call, z: dict(call=call, z=z)) if fancy else (lambda call, z: {})
        with torch.no_grad():
            m2, ls2 = self.actor_fn(self.P, s2, s.shape[0]) if fancy else self.actor_fn(self.P, s2)
            a2 = m2 + torch.exp(ls2) * eps_next
            tq = torch.min(self.critic_fn(self.TQ[0], s2, a2, **kw(1, 0)), self.critic_fn(self.TQ[1], s2, a2, **kw(1, 1)))
            y = r + GAMMA * (1 - d) * tq
        losses = []
        for i in range(2):
            q = self.critic_fn(self.Q[i], s, a, **kw(2, i))
            loss = F.mse_loss(q, y)
            grads = torch.autograd.grad(loss, list(self.Q[i].values()))
            self.opt_q[i].step(self.Q[i], dict(zip(self.Q[i].keys(), grads)))
            losses.append(float(loss.detach()))
        mean, ls = self.actor_fn(self.P, s)
        std = torch.exp(ls)
        a_new = mean + std * eps_new
        logp = (-((a_new - mean) ** 2) / (2 * std ** 2) - ls - math.log(math.sqrt(2 * math.pi))).sum(-1)
        qn = torch.min(self.critic_fn(self.Q[0], s, a_new, **kw(3, 0)), self.critic_fn(self.Q[1], s, a_new, **kw(3, 1)))
        ploss = -(qn - ALPHA * logp).mean()
        names = list(self.P.keys())
        grads = torch.autograd.grad(ploss, [self.P[k] for k in names], allow_unused=True)
        self.opt_p.step(self.P, dict(zip(names, grads)))
        with torch.no_grad():
            for i in range(2):
                for k in self.TQ[i]:
                    self.TQ[i][k].copy_(TAU * self.Q[i][k] + (1 - TAU) * self.TQ[i][k])
        self.updates += 1
        return losses[0], losses[1], float(ploss.detach())


# --- the BASELINE.json "256x256 MLP" family (legacy SACAgent shapes; not in the shipped reference code):
#     actor obs->256->256->2A ReLU, critics (obs+A)->256->256->1 ReLU, same update rule.
def mlp_actor_forward(P, obs):
    h = F.relu(obs @ P["0.weight"].T + P["0.bias"])
    h = F.relu(h @ P["2.weight"].T + P["2.bias"])
    out = h @ P["4.weight"].T + P["4.bias"]
    a_dim = out.shape[1] // 2
    return out[:, :a_dim], torch.clamp(out[:, a_dim:], -20, 2)


def mlp_critic_forward(Q, s, a):
    h = F.relu(torch.cat([s, a], -1) @ Q["0.weight"].T + Q["0.bias"])
    h = F.relu(h @ Q["2.weight"].T + Q["2.bias"])
    return (h @ Q["4.weight"].T + Q["4.bias"]).squeeze(-1)


# --- acting-path extras of the reference (both nets are created with random weights and never trained there)
def curiosity_reward(F_, prev_obs8, action, obs8):
    """CuriosityModule.compute_intrinsic_reward, env/enhanced_rocket_tvc_env.py:257-269, batched over rows:
    0.01 * mean((forward_model([prev_obs8 | action]) - obs8)^2).  F_: forward_model state_dict ('0','2','4')."""
    x = torch.cat([prev_obs8, action], -1)
    h = F.relu(x @ F_["0.weight"].T + F_["0.bias"])
    h = F.relu(h @ F_["2.weight"].T + F_["2.bias"])
    pred = h @ F_["4.weight"].T + F_["4.bias"]
    return 0.01 * ((pred - obs8) ** 2).mean(-1)


def safety_layer(S, state, proposed, max_tilt=0.52, max_w=5.0, max_effort=1.0):
    """SafetyLayer.forward (agent/multi_algorithm_agent.py:304-351) followed by get_action's clamp (:789)."""
    q, w = state[:, :4], state[:, 4:7]
    pitch = torch.asin(2 * (q[:, 3] * q[:, 1] - q[:, 2] * q[:, 0]))
    yaw = torch.atan2(2 * (q[:, 3] * q[:, 2] + q[:, 0] * q[:, 1]), 1 - 2 * (q[:, 1] ** 2 + q[:, 2] ** 2))
    tilt = torch.sqrt(pitch ** 2 + yaw ** 2)
    viol = (tilt > max_tilt) | (w.norm(dim=1) > max_w) | (proposed.norm(dim=1) > max_effort)
    x = torch.cat([state, proposed], -1)
    h = F.relu(x @ S["0.weight"].T + S["0.bias"])
    h = F.relu(h @ S["2.weight"].T + S["2.bias"])
    corr = h @ S["4.weight"].T + S["4.bias"]
    out = torch.where(viol.unsqueeze(1), corr, proposed)
    return torch.clamp(out, -1.0, 1.0), viol


def goal_logits(H, state):
    """HierarchicalAgent.high_level_policy (agent/multi_algorithm_agent.py:366-374): Linear-GELU-LN-Linear-GELU-LN-Linear."""
    h = _ln(F.gelu(state @ H["0.weight"].T + H["0.bias"]), H, "2")
    h = _ln(F.gelu(h @ H["3.weight"].T + H["3.bias"]), H, "5")
    return h @ H["6.weight"].T + H["6.bias"]


def goal_from_uniform(logits, u):
    """categorical draw by inverse CDF over softmax(logits) with u in [0,1): the role of torch.multinomial in select_goal (:396-402)"""
    p = torch.softmax(logits, dim=-1)
    cdf = torch.cumsum(p, dim=-1)
    return torch.clamp((u.unsqueeze(1) >= cdf).sum(dim=1), max=logits.shape[1] - 1)


def hierarchical_act(H, Plow, state, goal_idx, batch_pe=False):
    """HierarchicalAgent.get_action (:404-417): low-level policy on [state | one_hot(goal)] -> mean, log_std"""
    onehot = F.one_hot(goal_idx.long(), num_classes=H["6.weight"].shape[0]).to(state)
    return actor_forward(Plow, torch.cat([state, onehot], dim=-1), batch_pe=batch_pe)

