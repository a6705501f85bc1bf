"""ORACLE (test infrastructure only — never imported by the product path).

numpy fp32 restatement of the pixel front end of the DDPG-backbone agents:
  utils.RandomShiftsAug   /root/reference/utils/utils.py:222-254  (replicate pad 4, integer shift through F.grid_sample)
  ddpg.Encoder            /root/reference/agents/unsupervised_learning/ddpg.py:12-39 (obs/255 - 0.5, 4 x [Conv3x3 + ReLU], flatten)
Third-party arithmetic restated: torch.linspace (fp32 CPU: symmetric start/end formula), F.grid_sample bilinear with
align_corners=False and zero padding (unnormalise ((g + 1) * size - 1) / 2, four taps nw/ne/sw/se), nn.Conv2d cross-correlation.
Pinned by tests/golden/pixels_g5.npz (reference outputs).
"""
import numpy as np

F32 = np.float32


def _linspace_f32(start, end, steps):
    """torch.linspace(start, end, steps, dtype=float32) on CPU: step = (end - start)/(steps - 1) in fp32, first half counted from
    start, second half from end."""
    start, end = F32(start), F32(end)
    step = F32((end - start) / F32(steps - 1))
    half = steps // 2
    out = np.empty(steps, F32)
    for i in range(steps):
        out[i] = F32(start + step * F32(i)) if i < half else F32(end - step * F32(steps - i - 1))
    return out


def random_shifts_aug(x, shifts, pad=4):
    """x: (n, c, h, h) uint8 or float; shifts: (n, 2) integers in [0, 2 pad] = (x shift, y shift). Returns float32 (n, c, h, h)."""
    x = np.asarray(x).astype(F32)
    n, c, h, w = x.shape
    P = h + 2 * pad
    xp = np.pad(x, ((0, 0), (0, 0), (pad, pad), (pad, pad)), mode='edge')
    eps = 1.0 / P
    ar = _linspace_f32(-1.0 + eps, 1.0 - eps, P)[:h]
    out = np.zeros((n, c, h, w), F32)
    for b in range(n):
        sh = (np.asarray(shifts[b], F32) * F32(2.0 / P)).astype(F32)
        gx = (ar + sh[0]).astype(F32)                        # along the width
        gy = (ar + sh[1]).astype(F32)                        # along the height
        # grid_sampler_unnormalize, align_corners=False. In exact arithmetic ix = j + shift; in fp32 it lands within ~1e-5 of the
        # integer, so the bilinear tap mixes in ~1e-5 of a neighbouring pixel: outputs deviate from a pure integer shift by up to
        # ~2e-3 (pixel units), and two fp32 evaluation orders of this line differ from each other by as much.
        ix = (((gx + F32(1)) * F32(P) - F32(1)) / F32(2)).astype(F32)
        iy = (((gy + F32(1)) * F32(P) - F32(1)) / F32(2)).astype(F32)
        x0, y0 = np.floor(ix).astype(np.int64), np.floor(iy).astype(np.int64)
        wx1, wy1 = (ix - x0).astype(F32), (iy - y0).astype(F32)
        wx0, wy0 = (F32(1) - wx1).astype(F32), (F32(1) - wy1).astype(F32)

        def tap(yy, xx):
            ok = ((yy >= 0) & (yy < P))[:, None] & ((xx >= 0) & (xx < P))[None, :]
            v = xp[b][:, np.clip(yy, 0, P - 1)][:, :, np.clip(xx, 0, P - 1)]
            return np.where(ok[None], v, F32(0))
        out[b] = (tap(y0, x0) * (wy0[:, None] * wx0[None, :]) + tap(y0, x0 + 1) * (wy0[:, None] * wx1[None, :]) +
                  tap(y0 + 1, x0) * (wy1[:, None] * wx0[None, :]) + tap(y0 + 1, x0 + 1) * (wy1[:, None] * wx1[None, :])).astype(F32)
    return out


def _im2col(x, k, stride):
    n, c, h, w = x.shape
    oh, ow = (h - k) // stride + 1, (w - k) // stride + 1
    cols = np.empty((n, oh, ow, c, k, k), x.dtype)
    for ky in range(k):
        for kx in range(k):
            cols[:, :, :, :, ky, kx] = x[:, :, ky:ky + stride * oh:stride, kx:kx + stride * ow:stride].transpose(0, 2, 3, 1)
    return cols.reshape(n * oh * ow, c * k * k), oh, ow


def conv_fwd(x, W, b, stride):
    """nn.Conv2d(ci, co, 3, stride): cross-correlation, no padding. x (n, ci, h, w) -> (n, co, oh, ow)."""
    n = x.shape[0]
    co, ci, k, _ = W.shape
    cols, oh, ow = _im2col(x, k, stride)
    y = (cols @ W.reshape(co, -1).T + b).astype(F32)
    return y.reshape(n, oh, ow, co).transpose(0, 3, 1, 2).copy(), cols


def conv_bwd(x_shape, cols, W, dy, stride, need_dx=True):
    n, ci, h, w = x_shape
    co, _, k, _ = W.shape
    oh, ow = dy.shape[2], dy.shape[3]
    d2 = dy.transpose(0, 2, 3, 1).reshape(-1, co)
    dW = (d2.T @ cols).reshape(W.shape).astype(F32)
    db = d2.sum(0).astype(F32)
    dx = None
    if need_dx:
        dcols = (d2 @ W.reshape(co, -1)).astype(F32).reshape(n, oh, ow, ci, k, k)
        dx = np.zeros(x_shape, F32)
        for ky in range(k):
            for kx in range(k):
                dx[:, :, ky:ky + stride * oh:stride, kx:kx + stride * ow:stride] += dcols[:, :, :, :, ky, kx].transpose(0, 3, 1, 2)
    return dW, db, dx


STRIDES = (2, 1, 1, 1)


def encoder_fwd(p, obs):
    """p = [W0, b0, W1, b1, W2, b2, W3, b3] (convnet.{0,2,4,6}); obs (n, c, 84, 84) pixel values. Returns (n, 39200) and a cache."""
    a = (np.asarray(obs).astype(F32) / F32(255.0) - F32(0.5)).astype(F32)
    cache = []
    for l in range(4):
        y, cols = conv_fwd(a, p[2 * l], p[2 * l + 1], STRIDES[l])
        a_out = np.maximum(y, F32(0))
        cache.append((a.shape, cols, a_out))
        a = a_out
    return a.reshape(a.shape[0], -1), cache


def encoder_bwd(p, cache, dh, need_dx=False):
    n = dh.shape[0]
    d = dh.reshape(cache[-1][2].shape).astype(F32)
    grads = [None] * 8
    dx = None
    for l in range(3, -1, -1):
        x_shape, cols, a_out = cache[l]
        d = (d * (a_out > 0)).astype(F32)
        dW, db, d = conv_bwd(x_shape, cols, p[2 * l], d, STRIDES[l], need_dx=(l > 0 or need_dx))
        grads[2 * l], grads[2 * l + 1] = dW, db
    if need_dx:
        dx = (d / F32(255.0)).astype(F32)
    return grads, dx


# ---------------------------------------------------------------------------------------------------
# DDPG on pixels (agents/unsupervised_learning/ddpg.py, obs_type == 'pixels')
# ---------------------------------------------------------------------------------------------------
ENC_KEYS = [f'convnet.{i}.{w}' for i in (0, 2, 4, 6) for w in ('weight', 'bias')]
PIX_ACTOR_KEYS = ['trunk.0.weight', 'trunk.0.bias', 'trunk.1.weight', 'trunk.1.bias'] + [f'policy.{i}.{w}' for i in (0, 2, 4) for w in ('weight', 'bias')]
PIX_CRITIC_KEYS = (['trunk.0.weight', 'trunk.0.bias', 'trunk.1.weight', 'trunk.1.bias'] +
                   [f'{q}.{i}.{w}' for q in ('Q1', 'Q2') for i in (0, 2, 4) for w in ('weight', 'bias')])


def pixel_param_shapes(c_in, A, F, H, R=39200):
    enc = []
    for l in range(4):
        enc += [(32, c_in if l == 0 else 32, 3, 3), (32,)]
    tr = [(F, R), (F,), (F,), (F,)]
    actor = tr + [(H, F), (H,), (H, H), (H,), (A, H), (A,)]
    q = [(H, F + A), (H,), (H, H), (H,), (1, H), (1,)]
    return list(zip(ENC_KEYS, enc)), list(zip(PIX_ACTOR_KEYS, actor)), list(zip(PIX_CRITIC_KEYS, tr + q + q))


class OraclePixelDDPG:
    """ddpg.py:126-328 with obs_type='pixels', meta_dim=0: aug_and_encode, update_critic (the encoder steps on the critic loss),
    update_actor on the detached encoding, soft update."""

    def __init__(self, enc, actor, critic, lr=1e-4, tau=0.01, stddev=0.2, clip=0.3, update_every_steps=2):
        from . import nets
        from .intr import mlp_bwd, mlp_fwd
        self.nets, self.mlp_fwd, self.mlp_bwd = nets, mlp_fwd, mlp_bwd
        self.enc = [np.array(p, F32) for p in enc]
        self.actor = [np.array(p, F32) for p in actor]
        self.critic = [np.array(p, F32) for p in critic]
        self.critic_target = [p.copy() for p in self.critic]
        self.enc_opt, self.actor_opt, self.critic_opt = nets.Adam(self.enc, lr), nets.Adam(self.actor, lr), nets.Adam(self.critic, lr)
        self.tau, self.std, self.clip, self.every = tau, stddev, clip, update_every_steps

    def _actor(self, feat):
        h, c1 = self.nets.Trunk.fwd(self.actor[0:4], feat)
        pre, acts = self.mlp_fwd(self.actor[4:10], h)
        return np.tanh(pre).astype(F32), (c1, acts)

    def _critic(self, p, feat, action):
        h, c1 = self.nets.Trunk.fwd(p[0:4], feat)
        x = np.concatenate([h, action], 1)
        q1, a1 = self.mlp_fwd(p[4:10], x)
        q2, a2 = self.mlp_fwd(p[10:16], x)
        return q1, q2, (c1, a1, a2)

    def _critic_bwd(self, p, cache, dq1, dq2, need_dfeat):
        c1, a1, a2 = cache
        F = p[0].shape[0]
        g1, dx1 = self.mlp_bwd(p[4:10], a1, dq1, need_dx=True)
        g2, dx2 = self.mlp_bwd(p[10:16], a2, dq2, need_dx=True)
        dx = (dx1 + dx2).astype(F32)
        gt, dfeat = self.nets.Trunk.bwd(p[0:4], c1, dx[:, :F], need_dfeat)
        return gt + g1 + g2, dfeat, dx[:, F:]

    def update(self, batch, step, shifts_obs, shifts_next, noise_c, noise_a, train_encoder=True):
        """train_encoder=False: the caller passed obs.detach() to update_critic (proto.py:190-193 and the other reward-free agents), so
        encoder_opt.step() finds no gradients and the encoder does not move."""
        if step % self.every != 0:
            return {}
        obs, action, reward, discount, next_obs = batch[:5]
        action, reward, discount = [np.asarray(x, F32) for x in (action, reward, discount)]
        B = obs.shape[0]
        fo, ecache = encoder_fwd(self.enc, random_shifts_aug(obs, shifts_obs))
        fn, _ = encoder_fwd(self.enc, random_shifts_aug(next_obs, shifts_next))
        nets = self.nets
        # update_critic (ddpg.py:240-268)
        mu_n, _ = self._actor(fn)
        na = nets.truncated_normal_sample(mu_n, noise_c, self.std, self.clip)
        tq1, tq2, _ = self._critic(self.critic_target, fn, na)
        y = (reward + discount * np.minimum(tq1, tq2)).astype(F32)
        q1, q2, cc = self._critic(self.critic, fo, action)
        e1, e2 = (q1 - y).astype(F32), (q2 - y).astype(F32)
        m = dict(batch_reward=float(reward.mean(dtype=F32)), critic_target_q=float(y.mean(dtype=F32)), critic_q1=float(q1.mean(dtype=F32)),
                 critic_q2=float(q2.mean(dtype=F32)), critic_loss=float(((e1 * e1).sum(dtype=F32) + (e2 * e2).sum(dtype=F32)) / F32(B)))
        gc, dfeat, _ = self._critic_bwd(self.critic, cc, (F32(2) * e1 / F32(B)).astype(F32), (F32(2) * e2 / F32(B)).astype(F32), train_encoder)
        self.last_critic_grads = gc
        self.critic_opt.step(self.critic, gc)
        if train_encoder:
            ge, _ = encoder_bwd(self.enc, ecache, dfeat)
            self.last_enc_grads = ge
            self.enc_opt.step(self.enc, ge)
        # update_actor (ddpg.py:270-292) on obs.detach()
        mu, (c1, acts) = self._actor(fo)
        a = nets.truncated_normal_sample(mu, noise_a, self.std, self.clip)
        q1, q2, cc = self._critic(self.critic, fo, a)
        w1 = np.where(q1 < q2, F32(1), np.where(q1 == q2, F32(0.5), F32(0)))
        dq = F32(-1.0) / F32(B)
        _, _, da = self._critic_bwd(self.critic, cc, (dq * w1).astype(F32), (dq * (F32(1) - w1)).astype(F32), False)
        dpre = (da * (F32(1) - mu * mu)).astype(F32)
        gp, dh = self.mlp_bwd(self.actor[4:10], acts, dpre, need_dx=True)
        gt, _ = nets.Trunk.bwd(self.actor[0:4], c1, dh, False)
        self.last_actor_grads = gt + gp
        self.actor_opt.step(self.actor, self.last_actor_grads)
        A = mu.shape[1]
        m.update(actor_loss=float(-np.minimum(q1, q2).mean(dtype=F32)), actor_ent=float(nets.normal_entropy(self.std) * A),
                 actor_logprob=float(nets.normal_log_prob(a, mu, self.std).sum(dtype=F32) / F32(B)))
        nets.soft_update(self.critic, self.critic_target, self.tau)
        return m


class OracleProtoPixels:
    """ProtoAgent.update with obs_type='pixels' (proto.py:159-207): augment once; update_proto through encoder(obs) with the encoder in
    proto_opt and encoder_target(next_obs) in the Sinkhorn branch; reward from encoder(next_obs) AFTER that step; DDPG update (the
    encoder also steps with encoder_opt on the critic loss); Polyak updates of encoder_target, predictor_target, critic_target."""

    def __init__(self, ddpg, proto, encoder_target_tau=0.05, lr=1e-4):
        self.ddpg, self.proto, self.ttau = ddpg, proto, encoder_target_tau
        self.enc_t = [p.copy() for p in ddpg.enc]
        self.proto_enc_opt = ddpg.nets.Adam(ddpg.enc, lr)            # the encoder's slots in proto_opt: a second Adam state

    def update(self, batch, step, shifts_obs, shifts_next, u_cat, noise_c, noise_a):
        if step % self.ddpg.every != 0:
            return {}
        obs, action, extr, discount, next_obs = batch[:5]
        ao, an = random_shifts_aug(obs, shifts_obs), random_shifts_aug(next_obs, shifts_next)
        fo, ecache = encoder_fwd(self.ddpg.enc, ao)
        ft, _ = encoder_fwd(self.enc_t, an)
        loss = self.proto.update(fo, ft)
        ge, _ = encoder_bwd(self.ddpg.enc, ecache, self.proto.last_dobs)
        self.proto_enc_opt.step(self.ddpg.enc, ge)
        fn, _ = encoder_fwd(self.ddpg.enc, an)
        intr = self.proto.reward(fn, u_cat)
        self.last_intr = intr
        m = self.ddpg.update((obs, action, intr, discount, next_obs), step, shifts_obs, shifts_next, noise_c, noise_a, train_encoder=False)
        self.ddpg.nets.soft_update(self.ddpg.enc, self.enc_t, self.ttau)
        self.proto.soft_update()
        m.update(repr_loss=loss, intr_reward=float(intr.mean(dtype=F32)), extr_reward=float(np.asarray(extr, F32).mean(dtype=F32)))
        return m
