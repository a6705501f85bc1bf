"""ORACLE (test infrastructure only — never imported by the product path).

numpy fp32 restatement of the reference's agent.update() step for TD3+BC, TD3, BC and the
DDPG (states) backbone:
  /root/reference/agents/offline_learning/td3_bc.py:119-189, td3.py:117-186, bc.py:78-110,
  /root/reference/agents/unsupervised_learning/ddpg.py:240-328.
Noise is an explicit input (two _standard_normal((B,A)) draws per update, critic-target
first, actor second). Pinned by tests/golden/tiny_*.npz and full_*.json (reference outputs).
"""
import numpy as np

from . import nets
from .nets import F32, ActorNet, TwinCritic, SharedTrunkCritic, Adam

ACTOR_KEYS_OFFLINE = ['policy.0.weight', 'policy.0.bias', 'policy.1.weight', 'policy.1.bias',
                      'policy.3.weight', 'policy.3.bias', 'policy.5.weight', 'policy.5.bias']
CRITIC_KEYS_OFFLINE = [f'{q}.{i}.{w}' for q in ('q1_net', 'q2_net') for i in (0, 1, 3, 5) for w in ('weight', 'bias')]
ACTOR_KEYS_DDPG = ['trunk.0.weight', 'trunk.0.bias', 'trunk.1.weight', 'trunk.1.bias',
                   'policy.0.weight', 'policy.0.bias', 'policy.2.weight', 'policy.2.bias']
CRITIC_KEYS_DDPG = (['trunk.0.weight', 'trunk.0.bias', 'trunk.1.weight', 'trunk.1.bias'] +
                    [f'{q}.{i}.{w}' for q in ('Q1', 'Q2') for i in (0, 2) for w in ('weight', 'bias')])


def param_shapes(kind, O, A, H, sf_dim=None):
    """(actor [(key, shape)], critic [(key, shape)] or None) in the reference's parameter order.
    sf_dim: APS's CriticSF (aps.py:12-60) — the DDPG critic with sf_dim-wide heads; O then includes the task."""
    tr = lambda i: [(H, i), (H,), (H,), (H,)]
    hd = lambda o: [(H, H), (H,), (o, H), (o,)]
    ddpg = kind in ('ddpg', 'aps')
    actor = list(zip(ACTOR_KEYS_DDPG if ddpg else ACTOR_KEYS_OFFLINE, tr(O) + hd(A)))
    if kind == 'bc':
        return actor, None
    if kind == 'cql':
        actor = list(zip(ACTOR_KEYS_OFFLINE, tr(O) + hd(2 * A)))
    if ddpg:
        critic = list(zip(CRITIC_KEYS_DDPG, tr(O + A) + hd(sf_dim or 1) + hd(sf_dim or 1)))
    else:
        critic = list(zip(CRITIC_KEYS_OFFLINE, tr(O + A) + hd(1) + tr(O + A) + hd(1)))
    return actor, critic


def _min_grad(q1, q2):
    """torch.min(a, b) backward: gradient to the smaller, split evenly on ties."""
    w1 = np.where(q1 < q2, F32(1), np.where(q1 == q2, F32(0.5), F32(0)))
    return w1, (F32(1) - w1)


def make_sf_critic(sf_dim):
    """CriticSF (aps.py:12-60) as a drop-in for SharedTrunkCritic: heads emit sf_dim successor features, Q = task . features with
    the task in the last sf_dim columns of obs (the agent feeds cat(obs, task), aps.py:236-238)."""
    class SFCritic:
        @staticmethod
        def fwd(p, obs, action):
            o1, o2, c = SharedTrunkCritic.fwd(p, obs, action)
            task = obs[:, -sf_dim:]
            q1 = (task * o1).sum(1, keepdims=True, dtype=F32)
            q2 = (task * o2).sum(1, keepdims=True, dtype=F32)
            return q1, q2, (c, task)

        @staticmethod
        def bwd(p, caches, dq1, dq2, need_dx, need_dw=True):
            c, task = caches
            return SharedTrunkCritic.bwd(p, c, (dq1 * task).astype(F32), (dq2 * task).astype(F32), need_dx, need_dw)
    return SFCritic


class OracleAgent:
    """kind in {'td3_bc','td3','bc','ddpg','crr','aps'}; params are lists of float32 arrays (reference order)."""

    def __init__(self, kind, actor_params, critic_params=None, lr=1e-4, tau=0.01, stddev_schedule='0.2',
                 stddev_clip=0.3, alpha=2.5, update_every_steps=2, world_size=1, allreduce=None,
                 num_value_samples=10, weight_func='indicator', sf_dim=None):
        """world_size > 1: this instance is one data-parallel rank holding a shard of the global batch; `allreduce`
        (list of float32 arrays -> summed in place across ranks) is called at the three points where the reference's
        single-process update needs a batch-global quantity: critic grads, sum|Q| (td3_bc.py:154),
        actor grads. Means are over batch*world_size, so the ranks end up with the large-batch update."""
        self.world_size, self.allreduce = world_size, allreduce
        self.num_value_samples, self.weight_func = num_value_samples, weight_func
        self.kind = kind
        self.actor = [np.array(p, F32) for p in actor_params]
        self.actor_opt = Adam(self.actor, lr)
        self.C = SharedTrunkCritic if kind == 'ddpg' else TwinCritic
        if kind == 'aps':                       # aps.py:268-320: DDPG's update_critic / update_actor with the task-conditioned critic
            self.C = make_sf_critic(sf_dim)
        if kind != 'bc':
            self.critic = [np.array(p, F32) for p in critic_params]
            self.critic_target = [p.copy() for p in self.critic]      # td3_bc.py:93
            self.critic_opt = Adam(self.critic, lr)
        self.tau, self.sched, self.clip, self.alpha = tau, stddev_schedule, stddev_clip, alpha
        self.update_every_steps = update_every_steps

    # td3_bc.py:119-143 / td3.py:117-141 / ddpg.py:240-268
    def update_critic(self, obs, action, reward, discount, next_obs, std, noise):
        mu_n, _ = ActorNet.fwd(self.actor, next_obs)
        next_action = nets.truncated_normal_sample(mu_n, noise, std, self.clip)
        tq1, tq2, _ = self.C.fwd(self.critic_target, next_obs, next_action)
        target_q = (reward + (discount * np.minimum(tq1, tq2)).astype(F32)).astype(F32)
        q1, q2, caches = self.C.fwd(self.critic, obs, action)
        B = F32(obs.shape[0] * self.world_size)
        e1, e2 = (q1 - target_q).astype(F32), (q2 - target_q).astype(F32)
        grads, _ = self.C.bwd(self.critic, caches, (F32(2) * e1 / B).astype(F32), (F32(2) * e2 / B).astype(F32),
                              need_dx=False)
        m = np.array([target_q.sum(dtype=F32), q1.sum(dtype=F32), q2.sum(dtype=F32),
                      (e1 * e1).sum(dtype=F32) + (e2 * e2).sum(dtype=F32)], F32) / B
        if self.world_size > 1:
            self.allreduce(grads + [m])
        self.critic_opt.step(self.critic, grads)
        return dict(critic_target_q=float(m[0]), critic_q1=float(m[1]), critic_q2=float(m[2]), critic_loss=float(m[3])), grads

    # td3_bc.py:145-166 / td3.py:143-163 / ddpg.py:270-292
    def update_actor(self, obs, action, std, noise):
        mu, cache = ActorNet.fwd(self.actor, obs)
        a = nets.truncated_normal_sample(mu, noise, std, self.clip)
        q1, q2, caches = self.C.fwd(self.critic, obs, a)
        q = np.minimum(q1, q2)
        B, A = obs.shape[0] * self.world_size, mu.shape[1]
        m = {}
        stats = np.array([np.abs(q).sum(dtype=F32), q.sum(dtype=F32), ((mu - action) ** 2).sum(dtype=F32)], F32)
        if self.world_size > 1:
            self.allreduce([stats])
        if self.kind == 'td3_bc':
            lmbda = F32(self.alpha) / (stats[0] / F32(B))
            loss = -lmbda * stats[1] / F32(B) + stats[2] / F32(B * A)
            dq = np.full_like(q, -lmbda / F32(B))
            dmu_extra = (F32(2) * (mu - action) / F32(B * A)).astype(F32)
        else:
            loss = -stats[1] / F32(B)
            dq = np.full_like(q, F32(-1.0) / F32(B))
            dmu_extra = 0
        w1, w2 = _min_grad(q1, q2)
        _, dx = self.C.bwd(self.critic, caches, (dq * w1).astype(F32), (dq * w2).astype(F32), need_dx=True)
        dmu = (dx[:, obs.shape[1]:] + dmu_extra).astype(F32)      # straight-through sample, utils.py:135-138
        grads = ActorNet.bwd(self.actor, cache, dmu)
        if self.world_size > 1:
            self.allreduce(grads)
        self.actor_opt.step(self.actor, grads)
        m['actor_loss'] = float(loss)
        m['actor_ent'] = float(nets.normal_entropy(std) * A)
        if self.kind in ('ddpg', 'aps'):
            lp = np.array([nets.normal_log_prob(a, mu, std).sum(dtype=F32) / F32(B)], F32)
            if self.world_size > 1:
                self.allreduce([lp])
            m['actor_logprob'] = float(lp[0])
        return m, grads

    # crr.py:121-142,170-196
    def update_actor_crr(self, obs, action, std, noise):
        n, (Bl, A) = self.num_value_samples, action.shape
        B = Bl * self.world_size
        mu, cache = ActorNet.fwd(self.actor, obs)
        obses = np.repeat(obs, n, axis=0)                       # 'b x -> (b n) x'
        acts = nets.truncated_normal_sample(np.repeat(mu, n, axis=0), noise, std, self.clip)
        q1r, q2r, _ = self.C.fwd(self.critic, obses, acts)
        V = np.minimum(q1r, q2r).reshape(Bl, n, 1).mean(1, dtype=F32)
        q1, q2, _ = self.C.fwd(self.critic, obs, action)
        adv = (np.minimum(q1, q2) - V).astype(F32)
        if self.weight_func == 'identity':
            w = adv
        elif self.weight_func == 'indicator':
            w = np.sign(np.maximum(adv, F32(0))).astype(F32)
        else:
            w = np.clip(np.exp(adv), F32(0), F32(20.0)).astype(F32)
        logp = nets.normal_log_prob(action, mu, std).sum(-1, keepdims=True)
        loss = np.array([-(logp * w).sum(dtype=F32) / F32(B)], F32)
        dmu = (-w * (action - mu) / (F32(std) * F32(std)) / F32(B)).astype(F32)
        grads = ActorNet.bwd(self.actor, cache, dmu)
        if self.world_size > 1:
            self.allreduce(grads + [loss])
        self.actor_opt.step(self.actor, grads)
        self.last_w = w
        return dict(actor_loss=float(loss[0]), actor_ent=float(nets.normal_entropy(std) * A)), grads

    # bc.py:78-95
    def update_bc(self, obs, action, std):
        mu, cache = ActorNet.fwd(self.actor, obs)
        B, A = mu.shape[0] * self.world_size, mu.shape[1]
        logp = nets.normal_log_prob(action, mu, std).sum(-1, keepdims=True)
        loss = np.array([(-logp).sum(dtype=F32) / F32(B)], F32)
        dmu = (-(action - mu) / (F32(std) * F32(std)) / F32(B)).astype(F32)
        grads = ActorNet.bwd(self.actor, cache, dmu)
        if self.world_size > 1:
            self.allreduce(grads + [loss])
        loss = loss[0]
        self.actor_opt.step(self.actor, grads)
        return dict(actor_loss=float(loss), actor_ent=float(nets.normal_entropy(std) * A)), grads

    # td3_bc.py:168-189 / bc.py:97-110 / ddpg.py:298-328
    def update(self, batch, step, noise_critic=None, noise_actor=None):
        if self.kind in ('ddpg', 'aps') and step % self.update_every_steps != 0:
            return {}
        obs, action, reward, discount, next_obs = [np.asarray(x, F32) for x in batch[:5]]
        std = nets.schedule(self.sched, step)
        br = np.array([reward.sum(dtype=F32) / F32(reward.shape[0] * self.world_size)], F32)
        if self.world_size > 1:
            self.allreduce([br])
        m = dict(batch_reward=float(br[0]))
        if self.kind == 'bc':
            mm, self.last_actor_grads = self.update_bc(obs, action, std)
            m.update(mm)
            return m
        mc, self.last_critic_grads = self.update_critic(obs, action, reward, discount, next_obs, std, noise_critic)
        m.update(mc)
        if self.kind == 'crr':
            ma, self.last_actor_grads = self.update_actor_crr(obs, action, std, noise_actor)
        else:
            ma, self.last_actor_grads = self.update_actor(obs, action, std, noise_actor)
        m.update(ma)
        nets.soft_update(self.critic, self.critic_target, self.tau)
        return m


# ---------------------------------------------------------------------------------------------------
# CQL (agents/offline_learning/cql.py:59-286): SAC-style tanh-Gaussian actor (2A outputs), TD loss + conservative
# penalty over 3*n_samples re-evaluations of the critic, entropy-tuned actor.
# ---------------------------------------------------------------------------------------------------
def _softplus(x):
    return np.logaddexp(F32(0), x).astype(F32)


def squashed_log_prob(x, mu, std):
    """TransformedDistribution(Normal(mu,std), TanhTransform).log_prob(tanh(x)) with x from the transform cache
    (utils.py:152-196): Normal.log_prob(x) - 2 (log 2 - x - softplus(-2x)), per element."""
    var = (std * std).astype(F32)
    base = (-((x - mu) ** 2) / (F32(2) * var) - np.log(std) - F32(np.log(np.sqrt(2 * np.pi)))).astype(F32)
    ladj = (F32(2.0) * (F32(np.log(2.0)) - x - _softplus(F32(-2.0) * x))).astype(F32)
    return (base - ladj).astype(F32)


def uniform_from_normal(z):
    """The fixtures route Tensor.uniform_(-1,1) through the normal noise stream: u = -1 + 2 Phi(z) (tools/gen_golden.py)."""
    from math import erf, sqrt
    u = 0.5 * (1.0 + np.vectorize(erf)(z.astype(np.float64) / sqrt(2.0)))
    return (-1.0 + 2.0 * u).astype(F32)


class OracleCQL:
    def __init__(self, actor_params, critic_params, lr=1e-4, tau=0.01, alpha=0.01, n_samples=3, world_size=1, allreduce=None,
                 use_critic_lagrange=False, target_cql_penalty=5.0):
        self.actor = [np.array(p, F32) for p in actor_params]
        self.critic = [np.array(p, F32) for p in critic_params]
        self.critic_target = [p.copy() for p in self.critic]
        self.actor_opt, self.critic_opt = Adam(self.actor, lr), Adam(self.critic, lr)
        self.log_actor_alpha = [np.zeros(1, F32)]
        self.actor_alpha_opt = Adam(self.log_actor_alpha, lr)
        self.tau, self.alpha, self.n = tau, alpha, n_samples
        self.world_size, self.allreduce = world_size, allreduce
        self.use_critic_lagrange, self.target_cql_penalty = use_critic_lagrange, target_cql_penalty
        self.log_critic_alpha = [np.zeros(1, F32)]                 # cql.py:103-105,111-112
        self.critic_alpha_opt = Adam(self.log_critic_alpha, lr)

    def policy(self, obs):
        raw, cache = ActorNet.fwd_raw(self.actor, obs)
        A = raw.shape[1] // 2
        mu = np.tanh(raw[:, :A]).astype(F32)
        ls = raw[:, A:]
        std = np.exp(np.clip(ls, F32(-10), F32(2))).astype(F32)
        return mu, std, ls, cache

    def update(self, batch, step, z_next, z_rand, z_cur, z_nxt, z_actor, rand_is_uniform=False):
        obs, action, reward, discount, next_obs = [np.asarray(x, F32) for x in batch[:5]]
        Bl, A = action.shape
        B, n = Bl * self.world_size, self.n
        m = dict(batch_reward=float(reward.mean(dtype=F32)))
        # ---- critic (cql.py:152-232)
        mu_n, std_n, _, _ = self.policy(next_obs)
        mu_c, std_c, _, _ = self.policy(obs)
        next_action = np.tanh(mu_n + std_n * z_next).astype(F32)
        tq1, tq2, _ = TwinCritic.fwd(self.critic_target, next_obs, next_action)
        y = (reward + discount * np.minimum(tq1, tq2)).astype(F32)
        rand = z_rand if rand_is_uniform else uniform_from_normal(z_rand)
        acts = np.concatenate([rand.reshape(n * Bl, A),
                               np.tanh(mu_c[None] + std_c[None] * z_cur).astype(F32).reshape(n * Bl, A),
                               np.tanh(mu_n[None] + std_n[None] * z_nxt).astype(F32).reshape(n * Bl, A),
                               action], 0)
        obs_all = np.concatenate([np.tile(obs, (3 * n, 1)), obs], 0)                 # repeat(n,1,1): sample-major rows
        q1a, q2a, caches = TwinCritic.fwd(self.critic, obs_all, acts)
        q1, q2 = q1a[3 * n * Bl:], q2a[3 * n * Bl:]
        ws, lse_sum = [], F32(0)
        for qa in (q1a, q2a):
            cat = qa.reshape(3 * n + 1, Bl, 1)
            mx = cat.max(0, keepdims=True)
            ex = np.exp(cat - mx).astype(F32)
            se = ex.sum(0, keepdims=True, dtype=F32)
            lse_sum = lse_sum + (np.log(se) + mx).sum(dtype=F32) / F32(B)
            ws.append((ex / se).astype(F32))                                           # d logsumexp / d q
        penalty = lse_sum - (q1 + q2).sum(dtype=F32) / F32(B)
        alpha_c = F32(self.alpha)
        if self.use_critic_lagrange:                                                   # cql.py:201-213 (single process only)
            ea = np.exp(self.log_critic_alpha[0]).astype(F32)
            inside = (ea >= 0.0) & (ea <= 1000000.0)                                   # torch.clamp passes the gradient on [min, max]
            g = np.where(inside, F32(-0.5) * (penalty - F32(self.target_cql_penalty)) * ea, F32(0)).astype(F32)
            self.critic_alpha_opt.step(self.log_critic_alpha, [g])
            alpha_c = np.clip(np.exp(self.log_critic_alpha[0]).astype(F32), F32(0), F32(1000000.0))[0]
        dqs = []
        for w, q in zip(ws, (q1, q2)):
            d = (alpha_c * w / F32(B)).astype(F32)
            d[-1] += (F32(2) * (q - y) / F32(B) - alpha_c / F32(B)).astype(F32)
            dqs.append(d.reshape(-1, 1))
        e1, e2 = q1 - y, q2 - y
        mse = ((e1 * e1).sum(dtype=F32) + (e2 * e2).sum(dtype=F32)) / F32(B)
        grads, _ = TwinCritic.bwd(self.critic, caches, dqs[0], dqs[1], need_dx=False)
        if self.world_size > 1:
            self.allreduce(grads)
        self.critic_opt.step(self.critic, grads)
        m.update(critic_target_q=float(y.mean(dtype=F32)), critic_q1=float(q1.mean(dtype=F32)), critic_q2=float(q2.mean(dtype=F32)),
                 critic_loss=float(mse + alpha_c * penalty), critic_cql=float(penalty), critic_cql_logsum=float(lse_sum))
        self.last_critic_grads = grads
        # ---- actor (cql.py:234-263)
        mu, std, ls, cache = self.policy(obs)
        x = (mu + std * z_actor).astype(F32)
        yact = np.tanh(x).astype(F32)
        log_pi = squashed_log_prob(x, mu, std)
        target_entropy = F32(-A)
        s_lp = np.array([log_pi.sum(dtype=F32)], F32)
        if self.world_size > 1:
            self.allreduce([s_lp])
        mean_lp = s_lp[0] / F32(B * A)
        alpha_loss = -(self.log_actor_alpha[0][0] * (mean_lp + target_entropy))
        self.actor_alpha_opt.step(self.log_actor_alpha, [np.array([-(mean_lp + target_entropy)], F32)])
        alpha = np.exp(self.log_actor_alpha[0][0]).astype(F32)
        q1p, q2p, cc = TwinCritic.fwd(self.critic, obs, yact)
        w1, w2 = _min_grad(q1p, q2p)
        dq = np.full_like(q1p, F32(-1.0) / F32(B))
        _, dx = TwinCritic.bwd(self.critic, cc, (dq * w1).astype(F32), (dq * w2).astype(F32), need_dx=True)
        dy = dx[:, obs.shape[1]:]
        gx = (dy * (F32(1) - yact * yact)).astype(F32)                        # through y = tanh(x)
        c = alpha / F32(B * A)
        dmu = (gx + c * F32(2) * yact).astype(F32)                            # d log_pi / d mu = 2 tanh(x)
        dstd = (gx * z_actor + c * (F32(-1) / std + F32(2) * yact * z_actor)).astype(F32)
        draw_mu = (dmu * (F32(1) - mu * mu)).astype(F32)
        inside = ((ls >= F32(-10)) & (ls <= F32(2))).astype(F32)              # clamp passes gradient on [min, max]
        draw_ls = (dstd * std * inside).astype(F32)
        grads = ActorNet.bwd_raw(self.actor, cache, np.concatenate([draw_mu, draw_ls], 1))
        if self.world_size > 1:
            self.allreduce(grads)
        self.actor_opt.step(self.actor, grads)
        qmin = np.minimum(q1p, q2p)
        m.update(actor_loss=float(alpha * mean_lp - qmin.sum(dtype=F32) / F32(B)), actor_ent=float(-mean_lp),
                 actor_alpha=float(alpha), actor_alpha_loss=float(alpha_loss))
        self.last_actor_grads = grads
        nets.soft_update(self.critic, self.critic_target, self.tau)
        return m
