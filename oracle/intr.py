"""ORACLE (test infrastructure only — never imported by the product path).

numpy fp32 restatement of the intrinsic-reward modules that sit in front of the DDPG update in the
reference's unsupervised agents (states observations):
  RND      /root/reference/agents/unsupervised_learning/rnd.py:13-60 (module), :79-108 (update_rnd, compute_intr_reward)
  ICM      /root/reference/agents/unsupervised_learning/icm.py:12-45, :64-92
  ICM-APT  /root/reference/agents/unsupervised_learning/icm_apt.py:13-57, :86-110 (PBE reward: utils.py:279-319)
  Disagreement /root/reference/agents/unsupervised_learning/disagreement.py:11-47, :64-90
  DIAYN    /root/reference/agents/unsupervised_learning/diayn.py:15-29, :78-127
and of the update() that wires them into DDPG (rnd.py:110-159, icm.py:94-139, icm_apt.py:112-158,
disagreement.py:92-136, diayn.py:129-176).
Pinned by tests/golden/tiny_{rnd,icm,icm_apt,icm_apt-kth,disagreement,diayn}.npz (reference outputs).
"""
import numpy as np

from . import nets
from .knn import PBE, RMS
from .nets import F32, Adam, linear_bwd, linear_fwd

BN_EPS = F32(1e-5)           # nn.BatchNorm1d default


def mlp_fwd(p, x, final_tanh=False):
    """nn.Sequential(Linear, ReLU, Linear, ReLU, ..., Linear[, Tanh]); p = [W, b] * L."""
    acts = [x]
    n = len(p) // 2
    for l in range(n):
        z = linear_fwd(acts[-1], p[2 * l], p[2 * l + 1])
        if l < n - 1:
            z = np.maximum(z, F32(0))
        elif final_tanh:
            z = np.tanh(z).astype(F32)
        acts.append(z)
    return acts[-1], acts


def mlp_bwd(p, acts, dout, final_tanh=False, need_dx=False):
    n = len(p) // 2
    grads = [None] * len(p)
    d = dout
    for l in range(n - 1, -1, -1):
        a = acts[l + 1]
        if l < n - 1:
            d = (d * (a > 0)).astype(F32)
        elif final_tanh:
            d = (d * (F32(1) - a * a)).astype(F32)
        dW, db, d = linear_bwd(acts[l], p[2 * l], d, need_dx or l > 0)
        grads[2 * l], grads[2 * l + 1] = dW, db
    return grads, d


def l2_rows(x):
    """torch.norm(x, dim=-1, p=2, keepdim=True) and its (sub)gradient factor x/||x|| (0 where the norm is 0)."""
    n = np.sqrt((x * x).sum(-1, keepdims=True, dtype=F32)).astype(F32)
    with np.errstate(divide='ignore', invalid='ignore'):
        g = np.where(n > 0, x / n, F32(0)).astype(F32)
    return n, g


RND_KEYS = [f'{n}.{i}.{w}' for n in ('predictor', 'target') for i in (1, 3, 5) for w in ('weight', 'bias')]
ICM_KEYS = [f'{n}.{i}.{w}' for n in ('forward_net', 'backward_net') for i in (0, 2) for w in ('weight', 'bias')]
APT_KEYS = ['trunk.0.weight', 'trunk.0.bias', 'trunk.1.weight', 'trunk.1.bias'] + ICM_KEYS


DIS_KEYS = [f'ensemble.{m}.{i}.{w}' for m in range(5) for i in (0, 2) for w in ('weight', 'bias')]
DIAYN_KEYS = [f'skill_pred_net.{i}.{w}' for i in (0, 2, 4) for w in ('weight', 'bias')]
APS_KEYS = [f'state_feat_net.{i}.{w}' for i in (0, 2, 4) for w in ('weight', 'bias')]


def intr_param_shapes(kind, O, A, H, R):
    """[(key, shape)] in the module's parameters() order. R: rep_dim (RND / ICM-APT) or skill_dim (DIAYN)."""
    if kind == 'disagreement':
        return list(zip(DIS_KEYS, [(H, O + A), (H,), (O, H), (O,)] * 5))
    if kind == 'diayn':
        return list(zip(DIAYN_KEYS, [(H, O), (H,), (H, H), (H,), (R, H), (R,)]))
    if kind == 'aps':           # R = sf_dim
        return list(zip(APS_KEYS, [(H, O), (H,), (H, H), (H,), (R, H), (R,)]))
    if kind == 'rnd':
        one = [(H, O), (H,), (H, H), (H,), (R, H), (R,)]
        return list(zip(RND_KEYS, one + one))
    if kind == 'icm':
        return list(zip(ICM_KEYS, [(H, O + A), (H,), (O, H), (O,), (H, 2 * O), (H,), (A, H), (A,)]))
    return list(zip(APT_KEYS, [(R, O), (R,), (R,), (R,), (H, R + A), (H,), (R, H), (R,), (H, 2 * R), (H,), (A, H), (A,)]))


class OracleRND:
    """rnd.py: BatchNorm1d(affine=False, training) -> clamp(+-5) -> predictor / frozen target MLPs."""

    def __init__(self, params, lr=1e-4, scale=1.0, clip_val=5.0):
        self.p = [np.array(x, F32) for x in params]       # predictor (6) then target (6)
        self.opt = Adam(self.p[:6], lr)                    # target grads are None -> Adam skips them (rnd.py:41-42,76)
        self.scale, self.clip = scale, clip_val
        self.rms = RMS()
        O = self.p[0].shape[1]
        self.running_mean, self.running_var, self.num_batches = np.zeros(O, F32), np.ones(O, F32), 0

    def _norm(self, obs):
        B = obs.shape[0]
        mean = obs.mean(0, dtype=F32)
        var = ((obs - mean) ** 2).mean(0, dtype=F32)                       # biased, normalisation
        self.running_mean = (F32(0.9) * self.running_mean + F32(0.1) * mean).astype(F32)
        self.running_var = (F32(0.9) * self.running_var + F32(0.1) * var * F32(B / max(B - 1, 1))).astype(F32)
        self.num_batches += 1
        x = ((obs - mean) / np.sqrt(var + BN_EPS)).astype(F32)
        return np.clip(x, F32(-self.clip), F32(self.clip))

    def errors(self, obs):
        x = self._norm(obs)
        pred, acts = mlp_fwd(self.p[:6], x)
        targ, _ = mlp_fwd(self.p[6:], x)
        diff = (targ - pred).astype(F32)
        return (diff * diff).mean(-1, keepdims=True, dtype=F32), diff, acts

    def update(self, obs):                                                 # rnd.py:79-96
        err, diff, acts = self.errors(obs)
        B, R = diff.shape
        loss = err.mean(dtype=F32)
        dpred = (F32(-2.0) * diff / F32(R) / F32(B)).astype(F32)
        grads, _ = mlp_bwd(self.p[:6], acts, dpred)
        self.opt.step(self.p[:6], grads)
        self.last_grads = grads
        return float(loss)

    def reward(self, obs):                                                 # rnd.py:98-103
        err, _, _ = self.errors(obs)
        _, var = self.rms(err)
        return (F32(self.scale) * err / (np.sqrt(var) + F32(1e-8))).astype(F32)


class OracleICM:
    """icm.py: forward model on [obs|action] -> next_obs, inverse model on [obs|next_obs] -> tanh action."""

    def __init__(self, params, lr=1e-4, scale=1.0):
        self.p = [np.array(x, F32) for x in params]       # forward_net (4) then backward_net (4)
        self.opt = Adam(self.p, lr)
        self.scale = scale

    def errors(self, obs, action, next_obs):
        nhat, fa = mlp_fwd(self.p[:4], np.concatenate([obs, action], -1))
        ahat, ba = mlp_fwd(self.p[4:], np.concatenate([obs, next_obs], -1), final_tanh=True)
        fe, fg = l2_rows((next_obs - nhat).astype(F32))
        be, bg = l2_rows((action - ahat).astype(F32))
        return fe, be, fg, bg, fa, ba

    def update(self, obs, action, next_obs):                               # icm.py:64-84
        fe, be, fg, bg, fa, ba = self.errors(obs, action, next_obs)
        B = F32(obs.shape[0])
        loss = fe.mean(dtype=F32) + be.mean(dtype=F32)
        g1, _ = mlp_bwd(self.p[:4], fa, (-fg / B).astype(F32))
        g2, _ = mlp_bwd(self.p[4:], ba, (-bg / B).astype(F32), final_tanh=True)
        self.last_grads = g1 + g2
        self.opt.step(self.p, self.last_grads)
        return float(loss)

    def reward(self, obs, action, next_obs):                               # icm.py:86-92
        fe = self.errors(obs, action, next_obs)[0]
        return np.log(fe * F32(self.scale) + F32(1.0)).astype(F32)


class OracleICMAPT:
    """icm_apt.py: ICM in the space of a Linear-LayerNorm-Tanh trunk; reward = particle entropy of the trunk output."""

    def __init__(self, params, lr=1e-4, knn_rms=True, knn_k=12, knn_avg=True, knn_clip=0.0):
        self.p = [np.array(x, F32) for x in params]       # trunk (4), forward_net (4), backward_net (4)
        self.opt = Adam(self.p, lr)
        self.pbe = PBE(RMS(), knn_clip, knn_k, knn_avg, knn_rms)

    def update(self, obs, action, next_obs):                               # icm_apt.py:33-50,86-104
        B = obs.shape[0]
        rep, tc = nets.Trunk.fwd(self.p[:4], np.concatenate([obs, next_obs], 0))     # same weights on both
        ro, rn = rep[:B], rep[B:]
        nhat, fa = mlp_fwd(self.p[4:8], np.concatenate([ro, action], -1))
        ahat, ba = mlp_fwd(self.p[8:], np.concatenate([ro, rn], -1), final_tanh=True)
        fe, fg = l2_rows((rn - nhat).astype(F32))
        be, bg = l2_rows((action - ahat).astype(F32))
        loss = fe.mean(dtype=F32) + be.mean(dtype=F32)
        R = ro.shape[1]
        g1, dx1 = mlp_bwd(self.p[4:8], fa, (-fg / F32(B)).astype(F32), need_dx=True)
        g2, dx2 = mlp_bwd(self.p[8:], ba, (-bg / F32(B)).astype(F32), final_tanh=True, need_dx=True)
        d_ro = (dx1[:, :R] + dx2[:, :R]).astype(F32)
        d_rn = (fg / F32(B) + dx2[:, R:]).astype(F32)      # next_obs rep is also the forward model's regression target
        g0, _ = nets.Trunk.bwd(self.p[:4], tc, np.concatenate([d_ro, d_rn], 0), need_dx=False)
        self.last_grads = g0 + g1 + g2
        self.opt.step(self.p, self.last_grads)
        return float(loss)

    def reward(self, obs, action, next_obs):                               # icm_apt.py:106-110
        rep, _ = nets.Trunk.fwd(self.p[:4], obs)
        return self.pbe(rep).reshape(-1, 1)


class OracleDisagreement:
    """disagreement.py: ensemble of 5 forward models; loss = mean L2 error over (B, 5); reward = variance of the predictions."""

    def __init__(self, params, lr=1e-4, n_models=5):
        self.p = [np.array(x, F32) for x in params]
        self.opt = Adam(self.p, lr)
        self.n = n_models

    def update(self, obs, action, next_obs):                               # disagreement.py:19-33,64-80
        x = np.concatenate([obs, action], -1)
        B = obs.shape[0]
        grads, loss = [], F32(0)
        for m in range(self.n):
            p = self.p[4 * m:4 * m + 4]
            nhat, acts = mlp_fwd(p, x)
            e, g = l2_rows((next_obs - nhat).astype(F32))
            loss = loss + e.sum(dtype=F32)
            gm, _ = mlp_bwd(p, acts, (-g / F32(B * self.n)).astype(F32))
            grads += gm
        self.last_grads = grads
        self.opt.step(self.p, grads)
        return float(loss / F32(B * self.n))

    def reward(self, obs, action, next_obs):                               # disagreement.py:35-47,82-85
        x = np.concatenate([obs, action], -1)
        preds = np.stack([mlp_fwd(self.p[4 * m:4 * m + 4], x)[0] for m in range(self.n)], 0)
        return preds.var(0, ddof=1, dtype=F32).mean(-1, dtype=F32).astype(F32).reshape(-1, 1)


class OracleDIAYN:
    """diayn.py: skill discriminator on next_obs; cross-entropy loss; reward = log q(z|s') - log(1/skill_dim)."""

    def __init__(self, params, lr=1e-4, scale=1.0):
        self.p = [np.array(x, F32) for x in params]
        self.opt = Adam(self.p, lr)
        self.scale = scale

    def _logits(self, next_obs):
        d, acts = mlp_fwd(self.p, next_obs)
        mx = d.max(-1, keepdims=True)
        lse = (mx + np.log(np.exp(d - mx).sum(-1, keepdims=True, dtype=F32))).astype(F32)
        return d, (d - lse).astype(F32), acts

    def update(self, skill, next_obs):                                     # diayn.py:78-92,107-127
        B = next_obs.shape[0]
        z = skill.argmax(1)
        d, lsm, acts = self._logits(next_obs)
        loss = -lsm[np.arange(B), z].mean(dtype=F32)
        self.acc = float((lsm.argmax(1) == z).sum() / B)
        dd = np.exp(lsm).astype(F32)
        dd[np.arange(B), z] -= F32(1)
        grads, _ = mlp_bwd(self.p, acts, (dd / F32(B)).astype(F32))
        self.last_grads = grads
        self.opt.step(self.p, grads)
        return float(loss)

    def reward(self, skill, next_obs):                                     # diayn.py:94-105
        B, S = skill.shape
        _, lsm, _ = self._logits(next_obs)
        import math
        r = (lsm[np.arange(B), skill.argmax(1)] - F32(math.log(1 / S))).astype(F32).reshape(-1, 1)
        return (r * F32(self.scale)).astype(F32)


class OracleAPS:
    """aps.py:63-79,147-175: successor-feature net; loss = -mean(task . normalize(phi(s'))); reward = PBE(phi(s')) + task . phi/||phi||."""

    def __init__(self, params, lr=1e-4, knn_rms=True, knn_k=12, knn_avg=True, knn_clip=0.0001):
        self.p = [np.array(x, F32) for x in params]
        self.opt = Adam(self.p, lr)
        self.pbe = PBE(RMS(), knn_clip, knn_k, knn_avg, knn_rms)

    def update(self, task, next_obs):
        B = next_obs.shape[0]
        f, acts = mlp_fwd(self.p, next_obs)
        n = np.maximum(np.sqrt((f * f).sum(1, keepdims=True, dtype=F32)), F32(1e-12)).astype(F32)      # F.normalize
        fn = (f / n).astype(F32)
        loss = -(task * fn).sum(1, dtype=F32).mean(dtype=F32)
        dfn = (-task / F32(B)).astype(F32)
        df = ((dfn - fn * (fn * dfn).sum(1, keepdims=True, dtype=F32)) / n).astype(F32)
        grads, _ = mlp_bwd(self.p, acts, df)
        self.last_grads = grads
        self.opt.step(self.p, grads)
        return float(loss)

    def reward(self, task, next_obs):
        rep, _ = mlp_fwd(self.p, next_obs)
        ent = self.pbe(rep).reshape(-1, 1)
        repn = (rep / np.sqrt((rep * rep).sum(1, keepdims=True, dtype=F32))).astype(F32)      # torch.norm, no eps (aps.py:166)
        sf = (task * repn).sum(1, keepdims=True, dtype=F32)
        self.last_ent, self.last_sf = ent, sf
        return (ent + sf).astype(F32)


class OracleUnsupAgent:
    """{RND,ICM,ICMAPT}Agent.update with reward_free=True: module step, intrinsic reward, then the DDPG update on it."""

    def __init__(self, kind, ddpg, module):
        self.kind, self.ddpg, self.module = kind, ddpg, module

    def update(self, batch, step, noise_critic, noise_actor):
        if step % self.ddpg.update_every_steps != 0:
            return {}
        obs, action, extr, discount, next_obs = [np.asarray(x, F32) for x in batch[:5]]
        args = (obs,) if self.kind == 'rnd' else (obs, action, next_obs)
        if self.kind in ('diayn', 'aps'):
            skill = np.asarray(batch[5], F32)
            args = (skill, next_obs)
        loss = self.module.update(*args)
        intr = self.module.reward(*args)
        self.last_intr = intr
        if self.kind in ('diayn', 'aps'):  # the actor/critic see [obs | skill] (diayn.py:162-164) / [obs | task] (aps.py:236-238)
            obs, next_obs = np.concatenate([obs, skill], 1), np.concatenate([next_obs, skill], 1)
        m = self.ddpg.update((obs, action, intr, discount, next_obs), step, noise_critic, noise_actor)
        m[{'rnd': 'rnd_loss', 'disagreement': 'disagreement_loss', 'diayn': 'diayn_loss', 'aps': 'aps_loss'}.get(self.kind, 'icm_loss')] = loss
        if self.kind == 'aps':
            m['intr_ent_reward'] = float(self.module.last_ent.mean(dtype=F32))
            m['intr_sf_reward'] = float(self.module.last_sf.mean(dtype=F32))
        if self.kind == 'diayn':
            m['diayn_acc'] = self.module.acc
        m['intr_reward'] = float(intr.mean(dtype=F32))
        m['extr_reward'] = float(extr.mean(dtype=F32))
        if self.kind == 'rnd':
            m['pred_error_mean'] = float(self.module.rms.M[0])
            m['pred_error_std'] = float(np.sqrt(self.module.rms.S[0]))
        return m


# ---------------------------------------------------------------------------------------------------
# SMM (agents/unsupervised_learning/smm.py): VAE density model of [obs | z], skill discriminator, reward from their losses.
# ---------------------------------------------------------------------------------------------------
SMM_KEYS = ([f'z_pred_net.{i}.{w}' for i in (0, 2, 4) for w in ('weight', 'bias')] +
            [f'vae.{n}.{w}' for n in ('enc.0', 'enc.2', 'enc_mu', 'enc_logvar', 'dec.0', 'dec.2', 'dec.4') for w in ('weight', 'bias')])


def smm_param_shapes(O, Z, H, code_dim=128, vae_hidden=150):
    W = O + Z
    zp = [(H, O), (H,), (H, H), (H,), (Z, H), (Z,)]
    vae = [(vae_hidden, W), (vae_hidden,), (vae_hidden, vae_hidden), (vae_hidden,), (code_dim, vae_hidden), (code_dim,),
           (code_dim, vae_hidden), (code_dim,), (vae_hidden, code_dim), (vae_hidden,), (vae_hidden, vae_hidden), (vae_hidden,), (W, vae_hidden), (W,)]
    return list(zip(SMM_KEYS, zp + vae))


class OracleSMM:
    """smm.py:27-70 (VAE), :89-112 (SMM), :173-200 (update_vae / update_pred), :217-246 (reward)."""

    def __init__(self, params, sp_lr=1e-3, vae_lr=1e-2, vae_beta=0.5, state_ent_coef=1.0, latent_ent_coef=1.0, latent_cond_ent_coef=1.0,
                 goal=(150.0, 75.0)):
        self.p = [np.array(x, F32) for x in params]
        self.zp, self.vae = self.p[:6], self.p[6:]
        self.zp_opt, self.vae_opt = Adam(self.zp, sp_lr), Adam(self.vae, vae_lr)
        self.beta, self.coefs, self.goal = vae_beta, (state_ent_coef, latent_ent_coef, latent_cond_ent_coef), goal

    def update_vae(self, x, eps):
        v = self.vae
        B, W = x.shape
        h2, acts = mlp_fwd(v[0:4] + [np.eye(v[2].shape[0], dtype=F32), np.zeros(v[2].shape[0], F32)], x)     # enc = Linear-ReLU-Linear-ReLU
        h1 = acts[1]
        h2 = acts[2]
        mu, lv = linear_fwd(h2, v[4], v[5]), linear_fwd(h2, v[6], v[7])
        std = np.exp(F32(0.5) * lv).astype(F32)
        code = (eps * std + mu).astype(F32)
        out, dacts = mlp_fwd(v[8:14], code)
        diff = (x - out).astype(F32)
        sq = (diff * diff).astype(F32)
        kle = (F32(-0.5) * (F32(1) + lv - mu * mu - np.exp(lv)).sum(1, dtype=F32)).mean(dtype=F32)
        loss = F32(self.beta) * kle + sq.mean(dtype=F32)
        h_s_z = sq.sum(1, keepdims=True, dtype=F32)
        dout = (F32(-2.0) * diff / F32(B * W)).astype(F32)
        gdec, dcode = mlp_bwd(v[8:14], dacts, dout, need_dx=True)
        dmu = (dcode + F32(self.beta) * mu / F32(B)).astype(F32)
        dlv = (dcode * eps * F32(0.5) * std + F32(self.beta) * F32(-0.5) * (F32(1) - np.exp(lv)) / F32(B)).astype(F32)
        gWmu, gbmu, dh_mu = linear_bwd(h2, v[4], dmu)
        gWlv, gblv, dh_lv = linear_bwd(h2, v[6], dlv)
        dh2 = ((dh_mu + dh_lv) * (h2 > 0)).astype(F32)
        gW2, gb2, dh1 = linear_bwd(h1, v[2], dh2)
        dh1 = (dh1 * (h1 > 0)).astype(F32)
        gW1, gb1, _ = linear_bwd(x, v[0], dh1, need_dx=False)
        self.last_vae_grads = [gW1, gb1, gW2, gb2, gWmu, gbmu, gWlv, gblv] + gdec
        self.vae_opt.step(self.vae, self.last_vae_grads)
        return float(loss), h_s_z

    def update_pred(self, obs, z):
        B = obs.shape[0]
        logits, acts = mlp_fwd(self.zp, obs)
        lab = z.argmax(1)
        m = logits.max(1, keepdims=True)
        lsm = (logits - (m + np.log(np.exp(logits - m).sum(1, keepdims=True, dtype=F32)))).astype(F32)
        h_z_s = (-lsm[np.arange(B), lab]).reshape(-1, 1).astype(F32)
        dd = np.exp(lsm).astype(F32)
        dd[np.arange(B), lab] -= F32(1)
        grads, _ = mlp_bwd(self.zp, acts, (dd / F32(B)).astype(F32))
        self.last_pred_grads = grads
        self.zp_opt.step(self.zp, grads)
        return float(h_z_s.mean(dtype=F32)), h_z_s

    def log_p_star(self, obs):
        d = np.sqrt(((obs[:, 0] - F32(self.goal[0])) ** 2 + (obs[:, 1] - F32(self.goal[1])) ** 2).astype(F32)).astype(F32)
        return np.log(np.where(d > 1.0, F32(1) / d, F32(1)).astype(F32)).astype(F32)


class OracleSMMAgent:
    """SMMAgent.update, reward_free=True, states (smm.py:217-281).

    The reference adds the 1-D `log_p_star` (B,) to (B,1) terms, so its reward — and with it the TD target — is a (B,B) matrix:
    mse_loss(Q (B,1), target (B,B)) then averages over all pairs. Per sample that is the TD loss against
    reward_i = rest_i + mean_j log_p_star_j, plus the constant var_j(log_p_star_j) per critic in the loss VALUE. The oracle (and the
    HIP path) implement exactly that, bug-compatible and tested against the reference's own trajectory."""

    def __init__(self, ddpg, smm):
        self.ddpg, self.module = ddpg, smm

    def update(self, batch, step, eps, noise_critic, noise_actor):
        if step % self.ddpg.update_every_steps != 0:
            return {}
        obs, action, extr, discount, next_obs, z = [np.asarray(x, F32) for x in batch[:6]]
        sm = self.module
        loss_vae, h_s_z = sm.update_vae(np.concatenate([obs, z], 1), eps)
        loss_pred, h_z_s = sm.update_pred(obs, z)
        sec, lec, lcec = sm.coefs
        lps = sm.log_p_star(obs)
        h_z = F32(np.log(z.shape[1]))
        rest = (F32(sec) * h_s_z + F32(lec) * h_z + F32(lcec) * h_z_s).astype(F32)
        lps_mean = lps.mean(dtype=F32)
        reward = (rest + lps_mean).astype(F32)
        self.last_intr = reward
        m = self.ddpg.update((np.concatenate([obs, z], 1), action, reward, discount, np.concatenate([next_obs, z], 1)), step, noise_critic,
                             noise_actor)
        m['critic_loss'] = m['critic_loss'] + 2.0 * float(((lps - lps_mean) ** 2).mean(dtype=F32))
        m.update(intr_reward=float(reward.mean(dtype=F32)), log_p_star=float(lps_mean), pred_log_ratios=float((F32(sec) * h_s_z).mean(dtype=F32)),
                 latent_ent_coef=float(F32(lec) * h_z), latent_cond_ent_coef=float((F32(lcec) * h_z_s).mean(dtype=F32)), loss_vae=loss_vae,
                 loss_pred=loss_pred, extr_reward=float(extr.mean(dtype=F32)))
        return m
