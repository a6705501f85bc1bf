"""ORACLE (test infrastructure only — never imported by the product path; tools/micro/pixel_bench.py's CPU leg times it).

A torch-CPU twin of the reference's Proto agent on pixel observations — BASELINE.json configs[3] — written with the library ops the
reference uses (nn.Conv2d / nn.Linear / nn.LayerNorm / F.grid_sample / torch.optim.Adam, autograd backward), so that it turns a batch-1024
update round in seconds where the numpy restatement (oracle/pixels.py + oracle/proto.py) needs a minute. What it follows:
  RandomShiftsAug              utils/utils.py:222-254   (replicate pad 4, linspace base grid, integer shift * 2/(h+8), bilinear grid_sample)
  Encoder                      agents/unsupervised_learning/ddpg.py:12-39
  Actor / Critic (pixels)      ddpg.py:42-123           (trunk Linear+LayerNorm+Tanh; the action joins AFTER the critic's trunk)
  update_critic / update_actor ddpg.py:240-292
  ProtoAgent                   proto.py:46-207          (normalize_protos, update_proto with Sinkhorn-Knopp :13-28, compute_intr_reward, update)
The random draws are inputs: augmentation shifts, Categorical uniforms (inverse CDF on the double cumulative sum, the convention of
tools/gen_golden.py), TruncatedNormal noise. Pinned to the reference's recorded trajectories tests/golden/pixel_proto.npz (miniature) and
tests/golden/config4_proto_b1024.npz (shipped sizes) by tests/test_oracle_pixels.py.
"""
import copy

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


def shift_aug(x, shifts, pad=4, dtype=torch.float32):
    """x: (n, c, h, h) uint8 / float tensor; shifts: (n, 2) integers in [0, 2 pad]. fp32 arithmetic of RandomShiftsAug with the draw given
    (dtype=torch.float64: the same formulas in double, for sensitivity studies)."""
    x = x.to(dtype)
    n, _, h, w = x.shape
    assert h == w
    x = F.pad(x, (pad,) * 4, 'replicate')
    eps = 1.0 / (h + 2 * pad)
    line = torch.linspace(-1.0 + eps, 1.0 - eps, h + 2 * pad, dtype=x.dtype)[:h]
    gx = line.view(1, h, 1).expand(h, h, 1)                 # varies along the width
    base = torch.cat([gx, gx.transpose(1, 0)], dim=2).unsqueeze(0).repeat(n, 1, 1, 1)
    sh = torch.as_tensor(np.asarray(shifts)).to(x.dtype).view(n, 1, 1, 2) * (2.0 / (h + 2 * pad))
    return F.grid_sample(x, base + sh, padding_mode='zeros', align_corners=False)


class _Encoder(nn.Module):
    def __init__(self, c_in):
        super().__init__()
        self.convnet = nn.Sequential(nn.Conv2d(c_in, 32, 3, stride=2), nn.ReLU(), nn.Conv2d(32, 32, 3, stride=1), nn.ReLU(),
                                     nn.Conv2d(32, 32, 3, stride=1), nn.ReLU(), nn.Conv2d(32, 32, 3, stride=1), nn.ReLU())

    def forward(self, obs):
        return self.convnet(obs / 255.0 - 0.5).flatten(1)


def _head(i, h, o):
    return nn.Sequential(nn.Linear(i, h), nn.ReLU(inplace=True), nn.Linear(h, h), nn.ReLU(inplace=True), nn.Linear(h, o))


class _Actor(nn.Module):
    def __init__(self, R, A, Fd, H):
        super().__init__()
        self.trunk = nn.Sequential(nn.Linear(R, Fd), nn.LayerNorm(Fd), nn.Tanh())
        self.policy = _head(Fd, H, A)

    def forward(self, feat):
        return torch.tanh(self.policy(self.trunk(feat)))


class _Critic(nn.Module):
    def __init__(self, R, A, Fd, H):
        super().__init__()
        self.trunk = nn.Sequential(nn.Linear(R, Fd), nn.LayerNorm(Fd), nn.Tanh())
        self.Q1, self.Q2 = _head(Fd + A, H, 1), _head(Fd + A, H, 1)

    def forward(self, feat, action):
        x = torch.cat([self.trunk(feat), action], dim=-1)
        return self.Q1(x), self.Q2(x)


class _Projector(nn.Module):
    def __init__(self, pd, pj):
        super().__init__()
        self.trunk = nn.Sequential(nn.Linear(pd, pj), nn.ReLU(), nn.Linear(pj, pd))

    def forward(self, x):
        return self.trunk(x)


def _sinkhorn(scores, iters=3):
    """proto.py:13-28 on a copy: global max shift, exp, three row / column normalisation rounds, final column normalisation."""
    Q = torch.exp(scores - scores.max()).T
    Q = Q / Q.sum()
    r = torch.ones(Q.shape[0], dtype=Q.dtype) / Q.shape[0]
    c = torch.ones(Q.shape[1], dtype=Q.dtype) / Q.shape[1]
    for _ in range(iters):
        Q = Q * (r / Q.sum(dim=1)).unsqueeze(1)
        Q = Q * (c / Q.sum(dim=0)).unsqueeze(0)
    return (Q / Q.sum(dim=0, keepdim=True)).T


def _trunc_sample(mu, z, std, clip):
    eps = torch.as_tensor(z).to(mu.dtype) * std
    if clip is not None:
        eps = eps.clamp(-clip, clip)
    x = mu + eps
    return x - x.detach() + x.detach().clamp(-1.0 + 1e-6, 1.0 - 1e-6)


def _polyak(net, target, tau):
    with torch.no_grad():
        for p, t in zip(net.parameters(), target.parameters()):
            t.copy_(tau * p + (1 - tau) * t)


class TorchTwinProtoPixels:
    def __init__(self, c_in, hw, A, feature_dim, hidden, pred_dim, proj_dim, num_protos, queue_size, lr=1e-4, tau=0.1, encoder_target_tau=0.05,
                 critic_target_tau=0.01, topk=3, stddev=0.2, stddev_clip=0.3, dtype=torch.float32):
        self.dtype = dtype
        e = (hw - 3) // 2 + 1 - 6
        R = 32 * e * e
        self.encoder, self.actor, self.critic = _Encoder(c_in), _Actor(R, A, feature_dim, hidden), _Critic(R, A, feature_dim, hidden)
        self.predictor, self.projector = nn.Linear(R, pred_dim), _Projector(pred_dim, proj_dim)
        self.protos = nn.Linear(pred_dim, num_protos, bias=False)
        self.queue, self.queue_ptr = torch.zeros(queue_size, pred_dim, dtype=dtype), 0
        self.tau, self.ett, self.ctt, self.topk, self.std, self.clip, self.lr, self.NP, self.A = tau, encoder_target_tau, critic_target_tau, topk, stddev, stddev_clip, lr, num_protos, A
        self._finish()

    def _finish(self):
        for m in (self.encoder, self.actor, self.critic, self.predictor, self.projector, self.protos):
            m.to(self.dtype)
        self.critic_target, self.encoder_target, self.predictor_target = (copy.deepcopy(m) for m in (self.critic, self.encoder, self.predictor))
        mk = lambda *mods: torch.optim.Adam([p for m in mods for p in m.parameters()], lr=self.lr)
        self.encoder_opt, self.actor_opt, self.critic_opt = mk(self.encoder), mk(self.actor), mk(self.critic)
        self.proto_opt = mk(self.encoder, self.predictor, self.projector, self.protos)

    def load(self, params):
        """params: dict module name -> dict of arrays under the reference's state_dict keys; targets are re-copied, optimisers re-created."""
        for nm, sd in params.items():
            mod = getattr(self, nm)
            ref = mod.state_dict()
            mod.load_state_dict({k: torch.as_tensor(np.asarray(v)).reshape(ref[k].shape).to(self.dtype) for k, v in sd.items()})
        self._finish()

    def _normalize_protos(self):
        with torch.no_grad():
            self.protos.weight.copy_(F.normalize(self.protos.weight.clone(), dim=1, p=2))

    def update(self, batch, shifts_obs, shifts_next, u_cat, noise_c, noise_a):
        obs, action, extr, discount, next_obs = (torch.as_tensor(np.asarray(x)) for x in batch[:5])
        action, extr, discount = action.to(self.dtype), extr.to(self.dtype), discount.to(self.dtype)
        B = obs.shape[0]
        m = {}
        with torch.no_grad():
            ao, an = shift_aug(obs, shifts_obs, dtype=self.dtype), shift_aug(next_obs, shifts_next, dtype=self.dtype)
        # ---- update_proto (proto.py:114-157)
        self._normalize_protos()
        s = F.normalize(self.projector(self.predictor(self.encoder(ao))), dim=1, p=2)
        log_p = F.log_softmax(self.protos(s) / self.tau, dim=1)
        with torch.no_grad():
            t = F.normalize(self.predictor_target(self.encoder_target(an)), dim=1, p=2)
            q_t = _sinkhorn(self.protos(t) / self.tau)
        loss = -(q_t * log_p).sum(dim=1).mean()
        m['repr_loss'] = loss.item()
        self.proto_opt.zero_grad(set_to_none=True)
        loss.backward()
        self.proto_opt.step()
        # ---- compute_intr_reward on next_obs (proto.py:98-112)
        self._normalize_protos()
        with torch.no_grad():
            z = F.normalize(self.predictor(self.encoder(an)), dim=1, p=2)
            prob = F.softmax(self.protos(z).T, dim=1)
            cdf = torch.cumsum(prob.double(), dim=1).numpy()
            u = np.asarray(u_cat, np.float64)
            cand = [min(int(np.searchsorted(cdf[i], u[i] * cdf[i, -1], side='right')), cdf.shape[1] - 1) for i in range(self.NP)]
            self.queue[self.queue_ptr:self.queue_ptr + self.NP] = z[torch.tensor(cand)]
            self.queue_ptr = (self.queue_ptr + self.NP) % self.queue.shape[0]
            d = torch.norm(z[:, None, :] - self.queue[None, :, :], dim=2, p=2)
            reward = torch.topk(d, self.topk, dim=1, largest=False)[0][:, -1:]
        self.last_intr = reward.numpy().copy()
        m.update(intr_reward=reward.mean().item(), extr_reward=extr.mean().item(), batch_reward=reward.mean().item())
        with torch.no_grad():
            fo, fn = self.encoder(ao), self.encoder(an)
        # ---- update_critic on detached encodings (ddpg.py:240-268; encoder_opt finds no gradients and does not move)
        with torch.no_grad():
            na = _trunc_sample(self.actor(fn), noise_c, self.std, self.clip)
            tq1, tq2 = self.critic_target(fn, na)
            target_q = reward + discount * torch.min(tq1, tq2)
        q1, q2 = self.critic(fo, action)
        closs = F.mse_loss(q1, target_q) + F.mse_loss(q2, target_q)
        m.update(critic_target_q=target_q.mean().item(), critic_q1=q1.mean().item(), critic_q2=q2.mean().item(), critic_loss=closs.item())
        self.critic_opt.zero_grad(set_to_none=True)
        closs.backward()
        self.critic_opt.step()
        # ---- update_actor (ddpg.py:270-292)
        mu = self.actor(fo)
        a = _trunc_sample(mu, noise_a, self.std, self.clip)
        q = torch.min(*self.critic(fo, a))
        aloss = -q.mean()
        self.actor_opt.zero_grad(set_to_none=True)
        aloss.backward()
        self.actor_opt.step()
        var = self.std ** 2
        logp = (-((a.detach() - mu.detach()) ** 2) / (2 * var) - np.log(self.std) - np.log(np.sqrt(2 * np.pi))).sum(-1, keepdim=True)
        m.update(actor_loss=aloss.item(), actor_logprob=logp.mean().item(), actor_ent=float((0.5 + 0.5 * np.log(2 * np.pi) + np.log(self.std)) * self.A))
        _polyak(self.encoder, self.encoder_target, self.ett)
        _polyak(self.predictor, self.predictor_target, self.ett)
        _polyak(self.critic, self.critic_target, self.ctt)
        return m
