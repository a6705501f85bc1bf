"""ORACLE (test infrastructure only — never imported by the product path).

numpy fp32 restatement of the Proto-RL agent on state observations:
  sinkhorn_knopp   /root/reference/agents/unsupervised_learning/proto.py:14-31
  Projector        proto.py:34-43            ProtoAgent.__init__ (predictor, projector, protos, queue) proto.py:46-85
  normalize_protos proto.py:98-101           compute_intr_reward proto.py:103-124     update_proto proto.py:126-157
  update           proto.py:159-207 (DDPG update on the kNN reward, predictor_target Polyak with encoder_target_tau)
Categorical sampling (proto.py:112) is an explicit input: uniforms u in [0,1), inverse CDF over the softmax row.
Pinned by tests/golden/tiny_proto.npz (reference outputs; gen_golden.py routes Categorical.sample to the same inverse CDF).
"""
import math

import numpy as np

from . import nets
from .intr import mlp_bwd, mlp_fwd
from .knn import pairwise_l2, topk_smallest
from .nets import F32, Adam, linear_bwd, linear_fwd

PROTO_KEYS = ['predictor.weight', 'predictor.bias', 'projector.trunk.0.weight', 'projector.trunk.0.bias',
              'projector.trunk.2.weight', 'projector.trunk.2.bias', 'protos.weight']


def proto_param_shapes(O, pred_dim, proj_dim, num_protos):
    """[(key, shape)] in proto_opt's parameter order (states: the encoder is Identity and contributes none)."""
    return list(zip(PROTO_KEYS, [(pred_dim, O), (pred_dim,), (proj_dim, pred_dim), (proj_dim,), (pred_dim, proj_dim), (pred_dim,),
                                 (num_protos, pred_dim)]))


def l2_normalize(x, eps=1e-12):
    """F.normalize(x, dim=1, p=2): x / max(||x||, eps)."""
    n = np.maximum(np.sqrt((x * x).sum(1, keepdims=True, dtype=F32)), F32(eps)).astype(F32)
    return (x / n).astype(F32), n


def l2_normalize_bwd(dy, y, n):
    return ((dy - y * (y * dy).sum(1, keepdims=True, dtype=F32)) / n).astype(F32)


def sinkhorn_knopp(S):
    """proto.py:14-31 on S = scores / tau of shape (B, P); returns (B, P)."""
    Q = np.exp(S - S.max()).astype(F32).T.copy()
    Q /= Q.sum(dtype=F32)
    r = (np.ones(Q.shape[0], F32) / F32(Q.shape[0])).astype(F32)
    c = (np.ones(Q.shape[1], F32) / F32(Q.shape[1])).astype(F32)
    for _ in range(3):
        u = (r / Q.sum(1, dtype=F32)).astype(F32)
        Q *= u[:, None]
        Q *= (c / Q.sum(0, dtype=F32)).astype(F32)[None, :]
    Q = (Q / Q.sum(0, keepdims=True, dtype=F32)).astype(F32)
    return Q.T


def log_softmax(x):
    m = x.max(1, keepdims=True)
    return (x - (m + np.log(np.exp(x - m).sum(1, keepdims=True, dtype=F32)))).astype(F32)


class OracleProto:
    def __init__(self, params, queue_size, tau=0.1, encoder_target_tau=0.05, topk=3, lr=1e-4):
        self.p = [np.array(x, F32) for x in params]       # PROTO_KEYS order
        self.opt = Adam(self.p, lr)
        self.pt = [self.p[0].copy(), self.p[1].copy()]     # predictor_target (deepcopy at construction, proto.py:59)
        self.tau, self.ttau, self.topk = tau, encoder_target_tau, topk
        self.queue = np.zeros((queue_size, self.p[0].shape[0]), F32)
        self.queue_ptr = 0

    def normalize_protos(self):
        self.p[6] = l2_normalize(self.p[6])[0]
        # keep the optimiser's view pointing at the same storage
        self.opt_params_sync()

    def opt_params_sync(self):
        pass    # Adam.step takes the parameter list explicitly

    def update(self, obs, next_obs):                                      # proto.py:126-157
        self.normalize_protos()
        B = obs.shape[0]
        C = self.p[6]
        z1 = linear_fwd(obs, self.p[0], self.p[1])
        s, acts = mlp_fwd(self.p[2:6], z1)
        sn, nrm = l2_normalize(s)
        scores_s = (sn @ C.T).astype(F32)
        logp = log_softmax((scores_s / F32(self.tau)).astype(F32))
        t = l2_normalize(linear_fwd(next_obs, self.pt[0], self.pt[1]))[0]
        q = sinkhorn_knopp(((t @ C.T).astype(F32) / F32(self.tau)).astype(F32))
        loss = -(q * logp).sum(1, dtype=F32).mean(dtype=F32)
        dlogits = ((np.exp(logp) * q.sum(1, keepdims=True, dtype=F32) - q) / F32(B)).astype(F32)
        dscores = (dlogits / F32(self.tau)).astype(F32)
        dC = (dscores.T @ sn).astype(F32)
        dsn = (dscores @ C).astype(F32)
        ds = l2_normalize_bwd(dsn, sn, nrm)
        gproj, dz1 = mlp_bwd(self.p[2:6], acts, ds, need_dx=True)
        dW, db, _ = linear_bwd(obs, self.p[0], dz1, need_dx=False)
        self.last_dobs = (dz1 @ self.p[0]).astype(F32)        # d(loss)/d(obs): continues into the encoder on pixels (proto.py:75-78,131)
        self.last_grads = [dW, db] + gproj + [dC]
        self.opt.step(self.p, self.last_grads)
        return float(loss)

    def reward(self, next_obs, u):                                        # proto.py:103-124
        self.normalize_protos()
        C = self.p[6]
        z = l2_normalize(linear_fwd(next_obs, self.p[0], self.p[1]))[0]
        scores = (z @ C.T).astype(F32).T                                  # (P, B)
        m = scores.max(1, keepdims=True)
        e = np.exp(scores - m).astype(F32)
        prob = (e / e.sum(1, keepdims=True, dtype=F32)).astype(F32)
        cdf = np.cumsum(prob.astype(np.float64), 1)
        cand = np.array([min(int(np.searchsorted(cdf[i], u[i] * cdf[i, -1], side='right')), cdf.shape[1] - 1) for i in range(len(u))])
        self.last_candidates = cand
        P = C.shape[0]
        self.queue[self.queue_ptr:self.queue_ptr + P] = z[cand]
        self.queue_ptr = (self.queue_ptr + P) % self.queue.shape[0]
        return topk_smallest(pairwise_l2(z, self.queue), self.topk)[:, -1:]

    def soft_update(self):                                                # proto.py:200-203
        nets.soft_update(self.p[:2], self.pt, self.ttau)


def uniform_from_normal(z):
    return (0.5 * (1.0 + np.vectorize(math.erf)(np.asarray(z, np.float64) / math.sqrt(2.0))))


class OracleProtoAgent:
    """ProtoAgent.update with reward_free=True on states (encoder = Identity): proto step, kNN reward on next_obs, DDPG update."""

    def __init__(self, ddpg, proto):
        self.ddpg, self.module = ddpg, proto

    def update(self, batch, step, u_cat, noise_critic, noise_actor):
        if step % self.ddpg.update_every_steps != 0:
            return {}
        obs, action, extr, discount, next_obs = [np.asarray(x, F32) for x in batch[:5]]
        loss = self.module.update(obs, next_obs)
        intr = self.module.reward(next_obs, u_cat)
        self.last_intr = intr
        m = self.ddpg.update((obs, action, intr, discount, next_obs), step, noise_critic, noise_actor)
        self.module.soft_update()
        m['repr_loss'] = loss
        m['intr_reward'] = float(intr.mean(dtype=F32))
        m['extr_reward'] = float(extr.mean(dtype=F32))
        return m
