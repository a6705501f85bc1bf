"""ORACLE (test infrastructure only — never imported by the product path; bench.py's cpu_baseline leg times it).

A torch-CPU twin of the reference's TD3+BC update (agents/offline_learning/td3_bc.py:12-56 nets, :119-166 update_critic /
update_actor, :168-189 update; utils.TruncatedNormal utils/utils.py:128-149; soft_update_params :44-47), written with the same
library ops the reference uses (nn.Linear / nn.LayerNorm / F.mse_loss / torch.optim.Adam, autograd backward) so that its speed
stands for "the reference's PyTorch-CPU path" on whatever host runs the benchmark — the reference's own files never leave the
build container. Pinned against the reference's recorded trajectory (tests/golden/full_td3_bc.json) by
tests/test_oracle_agents.py::test_torch_twin_matches_reference_trajectory.
"""
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


def _mlp(in_dim, hidden, out_dim, squash):
    layers = [nn.Linear(in_dim, hidden), nn.LayerNorm(hidden), nn.Tanh(), nn.Linear(hidden, hidden), nn.ReLU(inplace=True),
              nn.Linear(hidden, out_dim)]
    return nn.Sequential(*layers, *([nn.Tanh()] if squash else []))


class _Twin(nn.Module):
    def __init__(self, obs_dim, act_dim, hidden):
        super().__init__()
        self.q1_net, self.q2_net = _mlp(obs_dim + act_dim, hidden, 1, False), _mlp(obs_dim + act_dim, hidden, 1, False)

    def forward(self, obs, action):
        x = torch.cat([obs, action], dim=-1)
        return self.q1_net(x), self.q2_net(x)


def _sample(mu, z, std, clip):
    """TruncatedNormal(mu, std).sample(clip): clipped noise, value clamped to +-(1-1e-6), gradient passed straight through."""
    eps = z * std
    if clip is not None:
        eps = eps.clamp(-clip, clip)
    x = mu + eps
    return x - x.detach() + x.detach().clamp(-1.0 + 1e-6, 1.0 - 1e-6)


class TorchTwinTD3BC:
    def __init__(self, obs_dim, act_dim, hidden, lr=1e-4, tau=0.01, alpha=2.5, stddev=0.2, stddev_clip=0.3, dtype=torch.float32):
        self.actor = _mlp(obs_dim, hidden, act_dim, True).to(dtype)
        self.critic = _Twin(obs_dim, act_dim, hidden).to(dtype)
        self.critic_target = _Twin(obs_dim, act_dim, hidden).to(dtype)
        self.critic_target.load_state_dict(self.critic.state_dict())
        self.actor_opt = torch.optim.Adam(self.actor.parameters(), lr=lr)
        self.critic_opt = torch.optim.Adam(self.critic.parameters(), lr=lr)
        self.tau, self.alpha, self.stddev, self.clip, self.dtype = tau, alpha, stddev, stddev_clip, dtype

    def load(self, actor_params, critic_params):
        """Lists of arrays in parameters() order (policy.{0,1,3,5}; q1_net.{0,1,3,5}, q2_net.{0,1,3,5})."""
        with torch.no_grad():
            for p, w in zip(self.actor.parameters(), actor_params):
                p.copy_(torch.as_tensor(w).reshape(p.shape))
            for p, w in zip(self.critic.parameters(), critic_params):
                p.copy_(torch.as_tensor(w).reshape(p.shape))
        self.critic_target.load_state_dict(self.critic.state_dict())

    def update(self, batch, noise_critic, noise_actor):
        obs, action, reward, discount, next_obs = (torch.as_tensor(x).to(self.dtype) for x in batch)
        zc, za = torch.as_tensor(noise_critic).to(self.dtype), torch.as_tensor(noise_actor).to(self.dtype)
        m = {}
        with torch.no_grad():
            next_action = _sample(self.actor(next_obs), zc, self.stddev, self.clip)
            tq1, tq2 = self.critic_target(next_obs, next_action)
            target_q = reward + discount * torch.min(tq1, tq2)
        q1, q2 = self.critic(obs, action)
        critic_loss = F.mse_loss(q1, target_q) + F.mse_loss(q2, target_q)
        self.critic_opt.zero_grad(set_to_none=True)
        critic_loss.backward()
        self.critic_opt.step()
        m.update(critic_target_q=target_q.mean().item(), critic_q1=q1.mean().item(), critic_q2=q2.mean().item(),
                 critic_loss=critic_loss.item())
        mu = self.actor(obs)
        pi = _sample(mu, za, self.stddev, self.clip)
        q = torch.min(*self.critic(obs, pi))
        lam = self.alpha / q.abs().mean().detach()
        actor_loss = -lam * q.mean() + F.mse_loss(mu, action)
        self.actor_opt.zero_grad(set_to_none=True)
        actor_loss.backward()
        self.actor_opt.step()
        m.update(actor_loss=actor_loss.item(), batch_reward=reward.mean().item())
        with torch.no_grad():
            for p, t in zip(self.critic.parameters(), self.critic_target.parameters()):
                t.copy_(self.tau * p + (1 - self.tau) * t)
        return m


def usable_cores():
    """(threads to use, description): physical cores of one socket, capped by this process's affinity mask and cgroup CPU quota
    (a 16-CPU cgroup share on a 128-core host is 16 usable cores, however many the kernel lists)."""
    n_aff = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    quota = None
    try:
        q, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if q != 'max':
            quota = max(1, int(int(q) / int(period)))
    except Exception:
        pass
    model, cores_socket0 = 'unknown', set()
    try:
        phys = core = None
        for line in open('/proc/cpuinfo'):
            k, _, v = line.partition(':')
            k, v = k.strip(), v.strip()
            if k == 'model name':
                model = v
            elif k == 'physical id':
                phys = v
            elif k == 'core id':
                core = v
            elif not k and phys is not None:
                if phys == '0':
                    cores_socket0.add(core)
                phys = core = None
    except Exception:
        pass
    per_socket = len(cores_socket0) or n_aff
    n = max(1, min(x for x in (n_aff, quota, per_socket) if x))
    return n, {'cpu_model': model, 'physical_cores_socket0': per_socket, 'affinity_cpus': n_aff, 'cgroup_cpu_quota': quota}
