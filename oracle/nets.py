"""ORACLE (test infrastructure only — never imported by the product path).

numpy fp32 restatement of the layer arithmetic torch performs for the reference's actor/critic
MLPs: nn.Linear, nn.LayerNorm, Tanh, ReLU, TruncatedNormal, Adam,
soft update — forward and hand-derived backward. Reference call sites are cited per function.
Pinned by tests/golden/utils_g2.npz, tiny_*.npz, full_*.json (generated from the reference).
"""
import math

import numpy as np

F32 = np.float32
LN_EPS = F32(1e-5)          # nn.LayerNorm default, td3_bc.py:17


# ------------------------------------------------------------------------------------ layers
def linear_fwd(x, W, b):
    # nn.Linear: y = x W^T + b, W stored (out, in)
    return (x @ W.T + b).astype(F32)


def linear_bwd(x, W, dy, need_dx=True):
    dW = (dy.T @ x).astype(F32)
    db = dy.sum(0).astype(F32)
    dx = (dy @ W).astype(F32) if need_dx else None
    return dW, db, dx


def layernorm_fwd(z, g, beta):
    # nn.LayerNorm(H): biased variance, eps inside sqrt
    mean = z.mean(-1, keepdims=True, dtype=F32)
    zc = z - mean
    var = (zc * zc).mean(-1, keepdims=True, dtype=F32)
    rstd = (F32(1.0) / np.sqrt(var + LN_EPS)).astype(F32)
    xhat = (zc * rstd).astype(F32)
    return (xhat * g + beta).astype(F32), xhat, rstd


def layernorm_bwd(dy, xhat, rstd, g):
    dg = (dy * xhat).sum(0).astype(F32)
    dbeta = dy.sum(0).astype(F32)
    dxh = dy * g
    m1 = dxh.mean(-1, keepdims=True, dtype=F32)
    m2 = (dxh * xhat).mean(-1, keepdims=True, dtype=F32)
    dz = (rstd * (dxh - m1 - xhat * m2)).astype(F32)
    return dz, dg, dbeta


# ------------------------------------------------------------------------------------ MLP blocks
class Trunk:
    """Linear(in,H) -> LayerNorm(H) -> Tanh   (td3_bc.py:16-17,38-39; ddpg.py:48-49,91-93).
    params: [W0, b0, ln_g, ln_b] in the reference's registration order."""

    @staticmethod
    def fwd(p, x):
        z = linear_fwd(x, p[0], p[1])
        y, xhat, rstd = layernorm_fwd(z, p[2], p[3])
        h = np.tanh(y).astype(F32)
        return h, (x, xhat, rstd, h)

    @staticmethod
    def bwd(p, cache, dh, need_dx):
        x, xhat, rstd, h = cache
        dy = (dh * (F32(1.0) - h * h)).astype(F32)
        dz, dg, dbeta = layernorm_bwd(dy, xhat, rstd, p[2])
        dW, db, dx = linear_bwd(x, p[0], dz, need_dx)
        return [dW, db, dg, dbeta], dx


class Head:
    """Linear(H,H) -> ReLU -> Linear(H,out)   (td3_bc.py:18-20,40-41; ddpg.py:52-62,96-108).
    params: [W1, b1, W2, b2]."""

    @staticmethod
    def fwd(p, h):
        a = np.maximum(linear_fwd(h, p[0], p[1]), F32(0))
        out = linear_fwd(a, p[2], p[3])
        return out, (h, a)

    @staticmethod
    def bwd(p, cache, dout, need_dx=True):
        h, a = cache
        dW2, db2, da = linear_bwd(a, p[2], dout)
        da = (da * (a > 0)).astype(F32)
        dW1, db1, dh = linear_bwd(h, p[0], da, need_dx)
        return [dW1, db1, dW2, db2], dh


class ActorNet:
    """Offline Actor (td3_bc.py:12-30, bc.py:13-31, td3.py:12-30, crr.py:12-30) and the DDPG
    states Actor (ddpg.py:42-76): trunk + head + tanh. params: 8 tensors, reference order."""
    N_PARAMS = 8

    @staticmethod
    def fwd(p, obs):
        h, c1 = Trunk.fwd(p[0:4], obs)
        pre, c2 = Head.fwd(p[4:8], h)
        mu = np.tanh(pre).astype(F32)
        return mu, (c1, c2, mu)

    @staticmethod
    def bwd(p, cache, dmu):
        c1, c2, mu = cache
        dpre = (dmu * (F32(1.0) - mu * mu)).astype(F32)
        g_head, dh = Head.bwd(p[4:8], c2, dpre)
        g_trunk, _ = Trunk.bwd(p[0:4], c1, dh, need_dx=False)
        return g_trunk + g_head

    # raw head outputs, no final tanh (CQL's 2A-wide actor, cql.py:12-31)
    @staticmethod
    def fwd_raw(p, obs):
        h, c1 = Trunk.fwd(p[0:4], obs)
        pre, c2 = Head.fwd(p[4:8], h)
        return pre, (c1, c2)

    @staticmethod
    def bwd_raw(p, cache, dpre):
        c1, c2 = cache
        g_head, dh = Head.bwd(p[4:8], c2, dpre)
        g_trunk, _ = Trunk.bwd(p[0:4], c1, dh, need_dx=False)
        return g_trunk + g_head


class TwinCritic:
    """Offline Critic (td3_bc.py:33-56): two independent trunk+head nets on cat(obs, action).
    params: 16 tensors (q1_net.{0,1,3,5}.*, q2_net.{0,1,3,5}.*)."""
    N_PARAMS = 16

    @staticmethod
    def fwd(p, obs, action):
        x = np.concatenate([obs, action], -1)
        qs, caches = [], []
        for i in range(2):
            q = p[8 * i:8 * i + 8]
            h, c1 = Trunk.fwd(q[0:4], x)
            out, c2 = Head.fwd(q[4:8], h)
            qs.append(out)
            caches.append((c1, c2))
        return qs[0], qs[1], caches

    @staticmethod
    def bwd(p, caches, dq1, dq2, need_dx, need_dw=True):
        grads, dx = [], None
        for i, dq in enumerate((dq1, dq2)):
            q = p[8 * i:8 * i + 8]
            c1, c2 = caches[i]
            g_head, dh = Head.bwd(q[4:8], c2, dq)
            g_trunk, dxi = Trunk.bwd(q[0:4], c1, dh, need_dx)
            grads += g_trunk + g_head
            if need_dx:
                dx = dxi if dx is None else (dx + dxi).astype(F32)
        return grads, dx


class SharedTrunkCritic:
    """DDPG states Critic (ddpg.py:79-123): one trunk on cat(obs, action), heads Q1, Q2.
    params: 12 tensors (trunk.{0,1}.*, Q1.{0,2}.*, Q2.{0,2}.*)."""
    N_PARAMS = 12

    @staticmethod
    def fwd(p, obs, action):
        x = np.concatenate([obs, action], -1)
        h, c1 = Trunk.fwd(p[0:4], x)
        q1, c2 = Head.fwd(p[4:8], h)
        q2, c3 = Head.fwd(p[8:12], h)
        return q1, q2, (c1, c2, c3)

    @staticmethod
    def bwd(p, caches, dq1, dq2, need_dx, need_dw=True):
        c1, c2, c3 = caches
        g1, dh1 = Head.bwd(p[4:8], c2, dq1)
        g2, dh2 = Head.bwd(p[8:12], c3, dq2)
        g_trunk, dx = Trunk.bwd(p[0:4], c1, (dh1 + dh2).astype(F32), need_dx)
        return g_trunk + g1 + g2, dx


# ------------------------------------------------------------------------------------ distributions
def truncated_normal_sample(mu, noise, std, clip):
    """utils.TruncatedNormal.sample (utils.py:140-149). Straight-through: d out / d mu = 1."""
    eps = (noise * F32(std)).astype(F32)
    if clip is not None:
        eps = np.clip(eps, F32(-clip), F32(clip))
    x = (mu + eps).astype(F32)
    return np.clip(x, F32(-1.0 + 1e-6), F32(1.0 - 1e-6)).astype(F32)


def normal_log_prob(a, mu, std):
    # torch.distributions.Normal.log_prob with scale = ones*std
    var = F32(std) * F32(std)
    return (-((a - mu) ** 2) / (F32(2) * var) - F32(math.log(std)) - F32(math.log(math.sqrt(2 * math.pi)))).astype(F32)


def normal_entropy(std):
    return F32(0.5 + 0.5 * math.log(2 * math.pi) + math.log(std))


def schedule(schdl, step):
    """utils.schedule (utils.py:199-219)."""
    import re
    try:
        return float(schdl)
    except ValueError:
        m = re.match(r'linear\((.+),(.+),(.+)\)', schdl)
        if m:
            init, final, duration = [float(g) for g in m.groups()]
            mix = float(np.clip(step / duration, 0.0, 1.0))
            return (1.0 - mix) * init + mix * final
        m = re.match(r'step_linear\((.+),(.+),(.+),(.+),(.+)\)', schdl)
        if m:
            init, final1, duration1, final2, duration2 = [float(g) for g in m.groups()]
            if step <= duration1:
                mix = float(np.clip(step / duration1, 0.0, 1.0))
                return (1.0 - mix) * init + mix * final1
            mix = float(np.clip((step - duration1) / duration2, 0.0, 1.0))
            return (1.0 - mix) * final1 + mix * final2
    raise NotImplementedError(schdl)


# ------------------------------------------------------------------------------------ optimiser
class Adam:
    """torch.optim.Adam single-tensor CPU path with defaults (td3_bc.py:96-97)."""

    def __init__(self, params, lr=1e-4, b1=0.9, b2=0.999, eps=1e-8):
        self.lr, self.b1, self.b2, self.eps = lr, b1, b2, eps
        self.m = [np.zeros_like(p) for p in params]
        self.v = [np.zeros_like(p) for p in params]
        self.t = 0

    def step(self, params, grads):
        self.t += 1
        bc1 = 1.0 - self.b1 ** self.t
        bc2 = 1.0 - self.b2 ** self.t
        step_size = self.lr / bc1
        bc2_sqrt = math.sqrt(bc2)
        for p, g, m, v in zip(params, grads, self.m, self.v):
            m += ((g - m) * F32(1.0 - self.b1)).astype(F32)                       # exp_avg.lerp_
            v *= F32(self.b2)
            v += (F32(1.0 - self.b2) * g * g).astype(F32)                         # addcmul_
            denom = (np.sqrt(v) / F32(bc2_sqrt) + F32(self.eps)).astype(F32)
            p += ((F32(-step_size) * m) / denom).astype(F32)                      # addcdiv_: self + (value*t1)/t2


def soft_update(net, target, tau):
    """utils.soft_update_params (utils.py:44-47)."""
    for p, t in zip(net, target):
        t[...] = (F32(tau) * p + F32(1.0 - tau) * t).astype(F32)
