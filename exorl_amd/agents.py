"""Agent classes with the reference's constructor kwargs and act()/update() signatures, running on
libexorl_hip.so. Hydra drop-in: `agent._target_=exorl_amd.agents.TD3BCAgent`.

Reference classes mirrored (file:line in /root/reference):
  TD3BCAgent  agents/offline_learning/td3_bc.py:59-189      TD3Agent  agents/offline_learning/td3.py:59-186
  BCAgent     agents/offline_learning/bc.py:34-110           DDPGAgent agents/unsupervised_learning/ddpg.py:126-328 (states)
  CRRAgent    agents/offline_learning/crr.py:59-219          CQLAgent  agents/offline_learning/cql.py:59-286
  RNDAgent    agents/unsupervised_learning/rnd.py:63-159     ICMAgent  agents/unsupervised_learning/icm.py:48-139
  ICMAPTAgent agents/unsupervised_learning/icm_apt.py:60-158 DisagreementAgent agents/unsupervised_learning/disagreement.py:50-136
  DIAYNAgent  agents/unsupervised_learning/diayn.py:32-176     ProtoAgent agents/unsupervised_learning/proto.py:46-207 (states)
  DDPGAgent with obs_type='pixels': Encoder ddpg.py:12-39, pixel Actor/Critic :42-123, update :213-328 (RandomShiftsAug utils.py:222-254)
  APSAgent    agents/unsupervised_learning/aps.py:82-320       SMMAgent   agents/unsupervised_learning/smm.py:115-281 (states)

Python here is orchestration only: it builds the initial weights with torch's CPU RNG in the reference's
construction order (so a given torch.manual_seed yields the reference's initial parameters), hands batches
to the engine and — under torch.distributed — all-reduces the flat gradient buffers between update phases.
"""
import functools
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn

from . import _lib as L
from . import utils
from .engine import AgentEngine, IntrEngine, PixelEngine

_OFFLINE_ACTOR_KEYS = ['policy.0.weight', 'policy.0.bias', 'policy.1.weight', 'policy.1.bias',
                       'policy.3.weight', 'policy.3.bias', 'policy.5.weight', 'policy.5.bias']
_OFFLINE_CRITIC_KEYS = [f'{q}.{i}.{w}' for q in ('q1_net', 'q2_net') for i in (0, 1, 3, 5) for w in ('weight', 'bias')]
_DDPG_ACTOR_KEYS = ['trunk.0.weight', 'trunk.0.bias', 'trunk.1.weight', 'trunk.1.bias',
                    'policy.0.weight', 'policy.0.bias', 'policy.2.weight', 'policy.2.bias']
_DDPG_CRITIC_KEYS = (['trunk.0.weight', 'trunk.0.bias', 'trunk.1.weight', 'trunk.1.bias'] +
                     [f'{q}.{i}.{w}' for q in ('Q1', 'Q2') for i in (0, 2) for w in ('weight', 'bias')])


def _mlp_init(in_dim, hidden, out_dim, n_trunks, n_heads):
    """Initial tensors of one net, drawn from torch's CPU generator in the order the reference's constructors
    consume it: every nn.Linear is default-initialised at construction, then `apply(utils.weight_init)`
    re-draws each Linear weight orthogonally and zeroes its bias (utils.py:59-69)."""
    mods = []
    if n_trunks == n_heads:                       # [trunk_i, head_i] per net (offline Actor/Critic, DDPG Actor)
        for _ in range(n_trunks):
            mods.append([nn.Linear(in_dim, hidden), nn.LayerNorm(hidden), nn.Linear(hidden, hidden), nn.Linear(hidden, out_dim)])
    else:                                         # DDPG Critic: trunk, then Q1, Q2 (ddpg.py:91-111)
        mods.append([nn.Linear(in_dim, hidden), nn.LayerNorm(hidden)])
        for _ in range(n_heads):
            mods.append([nn.Linear(hidden, hidden), nn.Linear(hidden, out_dim)])
    out = []
    for group in mods:
        for m in group:
            if isinstance(m, nn.Linear):
                nn.init.orthogonal_(m.weight.data)
                m.bias.data.fill_(0.0)
            out += [m.weight.data, m.bias.data]
    return out


class NetView:
    """Stands where the reference has an nn.Module (agent.actor / .critic / .critic_target): parameters(),
    state_dict(), load_state_dict(), train() over torch views of the engine's device buffers."""

    def __init__(self, engine, net, keys, on_change=None):
        self._engine, self._net, self._keys, self._on_change = engine, net, keys, on_change
        self.training = True
        self._params = []
        n = engine.num_tensors(net)
        assert n == len(keys), (n, len(keys))
        for i in range(n):
            self._params.append(engine.tensor(net, i, L.T_PARAM))

    def parameters(self):
        return list(self._params)

    def named_parameters(self):
        return list(zip(self._keys, self._params))

    def grads(self):
        return [self._engine.tensor(self._net, i, L.T_GRAD) for i in range(len(self._keys))]

    def state_dict(self):
        return OrderedDict((k, p.detach().clone()) for k, p in zip(self._keys, self._params))

    def load_state_dict(self, sd, strict=True):
        missing = [k for k in self._keys if k not in sd]
        if strict and (missing or len(sd) != len(self._keys)):
            raise KeyError(f'state_dict mismatch: missing {missing}, unexpected {[k for k in sd if k not in self._keys]}')
        for k, p in zip(self._keys, self._params):
            if k in sd:
                p.copy_(torch.as_tensor(sd[k]).to(p.device, torch.float32).reshape(p.shape))
        if self._on_change:
            self._on_change()

    def train(self, mode=True):
        self.training = mode
        return self

    def eval(self):
        return self.train(False)


class _AgentBase:
    KIND = None

    # -- pickling (pretrain.py:293-300 torch.save's the whole agent; finetune.py:222-252 loads it back) ---------------
    def __init_subclass__(cls, **kw):
        super().__init_subclass__(**kw)
        orig = cls.__dict__.get('__init__')
        if orig is None:
            return

        @functools.wraps(orig)
        def init(self, *a, **k):
            if not hasattr(self, '_ctor'):          # outermost constructor call: remember how this agent was built
                self._ctor = (a, dict(k))
            orig(self, *a, **k)
        cls.__init__ = init

    def _engines(self):
        return [('agent', self.engine)] + ([('intr', self.intr)] if hasattr(self, 'intr') else [])

    def __getstate__(self):
        """Constructor arguments + every device buffer that defines the training state (parameters, Adam moments and step
        counts, running statistics) as CPU tensors; host attributes that are plain data ride along."""
        torch.cuda.synchronize()
        eng = self.engine
        if getattr(self, 'obs_type', 'states') == 'pixels':
            st = {'ctor': self._ctor, 'pixel': eng.export_state(), 'training': getattr(self, 'training', True)}
        else:
            st = {'ctor': self._ctor, 'flat': {}, 'opt_steps': eng.opt_steps(), 'training': getattr(self, 'training', True)}
            nets = [L.NET_ACTOR] + ([L.NET_CRITIC, L.NET_CRITIC_TARGET] if eng.has_critic else [])
            for net in nets:
                whats = [L.T_PARAM] if net == L.NET_CRITIC_TARGET else [L.T_PARAM, L.T_ADAM_M, L.T_ADAM_V]
                st['flat'][net] = {w: eng.flat(net, w).cpu() for w in whats}
            if self.KIND == 'cql':
                st['cql_alpha'] = eng.cql_alpha_state().tolist()
        if hasattr(self, 'intr'):
            it = self.intr
            st['intr'] = {'flat': {w: it.flat(w).cpu() for w in (L.T_PARAM, L.T_ADAM_M, L.T_ADAM_V)}, 'rms': it.rms_state(),
                          'bn': it.bn.cpu() if it.bn is not None else None, 'opt_steps': it.opt_steps(),
                          'queue': (it.queue.cpu(), it.queue_ptr()) if it.queue is not None else None, 'counter': it.counter()}
        skip = {'engine', 'intr', 'actor', 'critic', 'critic_target', 'rnd', 'icm', 'pbe', 'intrinsic_reward_rms', 'disagreement', 'diayn',
                'predictor', 'predictor_target', 'projector', 'protos', 'queue', 'encoder_target', 'cat_hook', 'aps', 'smm', 'eps_hook', 'aug', 'encoder',
                'noise_hook', 'shift_hook', '_slots', '_graph_iter', '_graph_stddev', '_ctor', 'rnd_target_encoder', '_dobs', '_dp'}
        st['attrs'] = {k: v for k, v in self.__dict__.items() if k not in skip and not k.startswith('_keep')}
        return st

    def __setstate__(self, st):
        a, k = st['ctor']
        rng = torch.get_rng_state()                 # construction draws initial weights; loading must not disturb the caller's stream
        self._ctor = (a, dict(k))
        type(self).__init__(self, *a, **k)
        torch.set_rng_state(rng)
        eng = self.engine
        if 'pixel' in st:
            eng.import_state(st['pixel'])
        else:
            for net, bufs in st['flat'].items():
                for w, t in bufs.items():
                    eng.flat(net, w).copy_(t)
            eng.set_opt_steps(*st['opt_steps'])
            if 'cql_alpha' in st:
                eng.set_cql_alpha_state(*st['cql_alpha'])
            eng.params_changed(sync_target=False)
        if 'intr' in st:
            it, si = self.intr, st['intr']
            for w, t in si['flat'].items():
                it.flat(w).copy_(t)
            it.set_rms_state(*si['rms'])
            if si['bn'] is not None:
                it.bn.copy_(si['bn'])
            it.set_opt_steps(si['opt_steps'])
            if si.get('queue') is not None:
                it.queue.copy_(si['queue'][0])
                it.queue_ptr(si['queue'][1])
            if 'counter' in si:
                it.counter(si['counter'])
        self.__dict__.update(st['attrs'])
        self.train(st['training'])

    def _build(self, obs_dim, action_dim, hidden_dim, batch_size, lr, tau, alpha, stddev_clip, device, precision, seed,
               **engine_kw):
        ddpg = self.KIND in ('ddpg', 'aps')
        ws = 1
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            ws = torch.distributed.get_world_size()
            # every rank draws its own rows of the global batch's noise: without this all ranks share one Philox stream and the global
            # batch repeats each noise block world_size times (the reference's single process draws B x A iid normals)
            seed = (seed + 0x9E3779B1 * torch.distributed.get_rank()) & 0x7FFFFFFFFFFFFFFF if ws > 1 else seed
        self.world_size = ws
        # initial weights first (CPU RNG order: actor, critic, critic_target — td3_bc.py:86-93)
        actor0 = _mlp_init(obs_dim, hidden_dim, 2 * action_dim if self.KIND == 'cql' else action_dim, 1, 1)
        critic0 = None
        if self.KIND != 'bc':
            nt = 1 if ddpg else 2
            critic0 = _mlp_init(obs_dim + action_dim, hidden_dim, 1, nt, 2)
            _mlp_init(obs_dim + action_dim, hidden_dim, 1, nt, 2)       # critic_target's draws, overwritten by the copy
            if self.KIND == 'aps':      # APSAgent builds DDPG's scalar critics first, then replaces both with CriticSF (aps.py:94-104)
                critic0 = _mlp_init(obs_dim + action_dim, hidden_dim, engine_kw['sf_dim'], 1, 2)
                _mlp_init(obs_dim + action_dim, hidden_dim, engine_kw['sf_dim'], 1, 2)
        self.engine = AgentEngine(self.KIND, obs_dim, action_dim, hidden_dim, batch_size, lr=lr, tau=tau, alpha=alpha,
                                  stddev_clip=stddev_clip, precision=precision, world_size=ws, seed=seed, device=device,
                                  **engine_kw)
        self.actor = NetView(self.engine, L.NET_ACTOR, _DDPG_ACTOR_KEYS if ddpg else _OFFLINE_ACTOR_KEYS, self.params_changed)
        for p, w in zip(self.actor.parameters(), actor0):
            p.copy_(w.reshape(p.shape))
        if critic0 is not None:
            keys = _DDPG_CRITIC_KEYS if ddpg else _OFFLINE_CRITIC_KEYS
            self.critic = NetView(self.engine, L.NET_CRITIC, keys, self.params_changed)
            self.critic_target = NetView(self.engine, L.NET_CRITIC_TARGET, keys, self.params_changed)
            for p, w in zip(self.critic.parameters(), critic0):
                p.copy_(w.reshape(p.shape))
        self.engine.params_changed(sync_target=True)                   # critic_target.load_state_dict(critic.state_dict())
        self.engine.set_metrics(bool(getattr(self, 'use_tb', False) or getattr(self, 'use_wandb', False)))
        if ws > 1:
            from .comm import native_comm
            comm = native_comm(self.engine.device)          # RCCL inside the library when torch.distributed runs on nccl
            if comm is not None:
                self.engine.set_comm(comm)
        self._slots = None
        self._graph_iter = None
        self._graph_stddev = None
        self.noise_hook = None      # tests: callable(shape) -> np.ndarray standing in for _standard_normal

    def params_changed(self):
        """Call after writing parameter tensors in place (copy_ on .parameters(), dist.broadcast, ...): the kernels
        read derived copies (transposed first-layer weights, bf16 hidden weights) that must be rebuilt.
        load_state_dict / init_from / utils.hard_update_params do it themselves."""
        self.engine.params_changed(sync_target=False)

    # -- nn.Module-ish surface used by utils.eval_mode and the training scripts
    def train(self, training=True):
        self.training = training
        self.actor.train(training)
        if hasattr(self, 'critic'):
            self.critic.train(training)

    def _stddev(self, step):
        return utils.schedule(self.stddev_schedule, step)

    # -- hipGraph fast path: sampler + whole update captured once, replayed with one launch per step
    def enable_graph(self, replay_iter, step=0):
        """Binds `replay_iter` (an ArenaIterator with the Philox sampler) and captures sample+update.
        Returns False (and stays eager) when the iterator cannot be captured. A data-parallel step is captured too when its all-reduces
        are the library's own (RCCL, `engine.comm`): they are enqueued on the capture stream between the phases like any kernel."""
        eng = getattr(replay_iter, 'engine', None)
        if eng is None or getattr(replay_iter, 'sampler', None) != L.SAMPLER_PHILOX:
            return False
        if self.world_size != 1 and self.engine.comm is None:       # collectives in Python between the phases: nothing to capture
            return False
        self._graph_stddev = self._stddev(step)
        ok = True
        try:
            self.engine.enable_graph(eng, replay_iter.nstep, replay_iter.discount, self._graph_stddev)
        except L.ExorlError:
            if self.world_size == 1:
                raise
            ok = False
        if self.world_size != 1:                 # a capture refused on one rank must send EVERY rank down the eager path
            flag = torch.tensor([1.0 if ok else 0.0], device=self.engine.device)
            torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)
            if float(flag[0]) == 0.0:
                if ok:
                    self.engine.disable_graph()
                return False
        self._graph_iter = replay_iter
        return True

    def disable_graph(self):
        if self._graph_iter is not None:
            self.engine.disable_graph()
        self._graph_iter = None

    def _step(self, replay_iter, stddev):
        if replay_iter is self._graph_iter and self.noise_hook is None:
            self.engine.step_graph(stddev)            # the std lives in device memory: a moving schedule needs no re-capture
            return
        self._load_batch(replay_iter)
        self._run_update(stddev)

    def _load_batch(self, replay_iter):
        if hasattr(replay_iter, 'sample_into'):                                         # HBM sampler: replay -> update, zero copy
            if self._slots is None:
                self._slots = self.engine.batch_slots()
            replay_iter.sample_into(self._slots, self.engine.batch)
        else:
            batch = next(replay_iter)                                                   # any iterator of 5-tuples
            self.engine.set_batch(*batch[:5])

    def _noise(self, rows=None):
        if self.noise_hook is None:
            return None
        return self.noise_hook((rows or self.engine.batch, self.action_dim))

    def _run_update(self, stddev):
        """One gradient step; under torch.distributed the three global quantities are sum-all-reduced."""
        eng = self.engine
        if self.KIND == 'bc':
            nc = na = None
        else:
            # reference draw order: critic target, then actor; CRR's second draw is (B*n, A) (crr.py:125)
            nc, na = self._noise(), self._noise(getattr(self, '_second_noise_rows', None))
        if self.world_size == 1 or eng.comm is not None:      # one host call: with a communicator the library all-reduces between its phases
            eng.update(stddev, nc, na)
            return
        dist = torch.distributed                              # gloo / EXORL_DP_COMM=torch: the collectives stay here
        eng.update_phase(0, stddev, nc, na)
        if eng.has_critic:
            dist.all_reduce(eng.flat(L.NET_CRITIC, L.T_GRAD))
        eng.update_phase(1, stddev, nc, na)
        if self.KIND == 'td3_bc':                    # only TD3+BC's lambda needs a batch-global statistic
            dist.all_reduce(eng.stats())
        eng.update_phase(2, stddev, nc, na)
        dist.all_reduce(eng.flat(L.NET_ACTOR, L.T_GRAD))
        eng.update_phase(3, stddev, nc, na)

    def _metrics(self, keys, stddev):
        raw = self.engine.metrics_raw()
        if self.world_size > 1:                          # partial means -> global means
            t = torch.from_numpy(raw.copy()).to(self.engine.device)
            torch.distributed.all_reduce(t)
            raw = t.cpu().numpy()
        m = {}
        for idx, name in keys:
            m[name] = float(raw[idx])
        # TruncatedNormal inherits Normal.entropy: 0.5 + 0.5 log(2 pi) + log(std), summed over action dims
        m['actor_ent'] = float(np.float32(0.5 + 0.5 * np.log(2 * np.pi) + np.log(stddev)) * self.action_dim)
        return m

    def _act(self, obs_vec, step, eval_mode):
        stddev = self._stddev(step)
        noise = self.noise_hook((1, self.action_dim)) if (self.noise_hook and not eval_mode) else None
        explore = (not eval_mode) and step < self.num_expl_steps
        if not explore and isinstance(obs_vec, np.ndarray):
            a = self.engine.act_host(obs_vec, stddev, eval_mode, noise)      # one kernel launch, result in a pinned host slot
            if a is not None:
                return a
        if eval_mode:
            a = self.engine.act(obs_vec, stddev, True)
        else:
            a = self.engine.act(obs_vec, stddev, False, noise)
            if explore:
                a.uniform_(-1.0, 1.0)
        return a.cpu().numpy()[0]


_CRITIC_METRICS = [(L.M_BATCH_REWARD, 'batch_reward'), (L.M_CRITIC_TARGET_Q, 'critic_target_q'), (L.M_CRITIC_Q1, 'critic_q1'),
                   (L.M_CRITIC_Q2, 'critic_q2'), (L.M_CRITIC_LOSS, 'critic_loss'), (L.M_ACTOR_LOSS, 'actor_loss')]


class TD3BCAgent(_AgentBase):
    KIND = 'td3_bc'

    def __init__(self, name, obs_shape, action_shape, device, lr, hidden_dim, critic_target_tau, stddev_schedule, nstep,
                 batch_size, stddev_clip, use_tb, alpha, has_next_action=False, *, precision='fp32', seed=0):
        self.action_dim = action_shape[0]
        self.hidden_dim = hidden_dim
        self.lr = lr
        self.device = device
        self.critic_target_tau = critic_target_tau
        self.use_tb = use_tb
        self.stddev_schedule = stddev_schedule
        self.stddev_clip = stddev_clip
        self.alpha = alpha
        self._build(obs_shape[0], action_shape[0], hidden_dim, batch_size, lr, critic_target_tau, alpha, stddev_clip, device,
                    precision, seed)
        self.train()
        self.critic_target.train()

    def act(self, obs, step, eval_mode):
        return self._act(np.asarray(obs, np.float32), step, eval_mode)

    def update(self, replay_iter, step):
        metrics = dict()
        stddev = self._stddev(step)
        self._step(replay_iter, stddev)
        if self.use_tb:
            metrics.update(self._metrics(_CRITIC_METRICS, stddev))
        return metrics


class TD3Agent(TD3BCAgent):
    KIND = 'td3'

    def __init__(self, name, obs_shape, action_shape, device, lr, hidden_dim, critic_target_tau, stddev_schedule, nstep,
                 batch_size, stddev_clip, use_tb, has_next_action=False, *, precision='fp32', seed=0):
        super().__init__(name, obs_shape, action_shape, device, lr, hidden_dim, critic_target_tau, stddev_schedule, nstep,
                         batch_size, stddev_clip, use_tb, 0.0, has_next_action, precision=precision, seed=seed)


class CRRAgent(_AgentBase):
    """agents/offline_learning/crr.py:59-219."""
    KIND = 'crr'

    def __init__(self, name, obs_shape, action_shape, device, lr, hidden_dim, critic_target_tau, num_value_samples, weight_func,
                 stddev_schedule, nstep, batch_size, stddev_clip, use_tb, has_next_action=False, *, precision='fp32', seed=0):
        assert weight_func in ['identity', 'indicator', 'exp']
        self.action_dim = action_shape[0]
        self.hidden_dim = hidden_dim
        self.lr = lr
        self.device = device
        self.critic_target_tau = critic_target_tau
        self.use_tb = use_tb
        self.stddev_schedule = stddev_schedule
        self.stddev_clip = stddev_clip
        self.num_value_samples = num_value_samples
        self.weight_func = weight_func
        self._build(obs_shape[0], action_shape[0], hidden_dim, batch_size, lr, critic_target_tau, 0.0, stddev_clip, device, precision,
                    seed, num_value_samples=num_value_samples, weight_func=weight_func)
        self._second_noise_rows = batch_size * num_value_samples
        self.train()
        self.critic_target.train()

    def act(self, obs, step, eval_mode):
        return self._act(np.asarray(obs, np.float32), step, eval_mode)

    def update(self, replay_iter, step):
        metrics = dict()
        stddev = self._stddev(step)
        self._step(replay_iter, stddev)
        if self.use_tb:
            metrics.update(self._metrics(_CRITIC_METRICS, stddev))
        return metrics


class CQLAgent(_AgentBase):
    """agents/offline_learning/cql.py:59-286 (both the shipped cql.yaml and use_critic_lagrange=True; under torch.distributed the latter splits
    phase 0 around a sum-all-reduce of the penalty, _run_update)."""
    KIND = 'cql'

    def __init__(self, name, obs_shape, action_shape, device, lr, hidden_dim, critic_target_tau, nstep, batch_size, use_tb, alpha,
                 n_samples, target_cql_penalty, use_critic_lagrange, has_next_action=False, *, precision='fp32', seed=0):
        self.action_dim = action_shape[0]
        self.hidden_dim = hidden_dim
        self.lr = lr
        self.device = device
        self.critic_target_tau = critic_target_tau
        self.use_tb = use_tb
        self.use_critic_lagrange = use_critic_lagrange
        self.target_cql_penalty = target_cql_penalty
        self.alpha = alpha
        self.n_samples = n_samples
        self.target_entropy = -self.action_dim
        self.stddev_schedule = '1.0'          # unused by CQL (state-dependent std); keeps the shared plumbing uniform
        self._build(obs_shape[0], action_shape[0], hidden_dim, batch_size, lr, critic_target_tau, alpha, 0.0, device, precision, seed,
                    n_samples=n_samples, use_critic_lagrange=use_critic_lagrange, target_cql_penalty=target_cql_penalty)
        self.train()
        self.critic_target.train()

    @property
    def log_actor_alpha(self):
        return torch.tensor([self.engine.cql_alpha_state()[0]])

    @property
    def log_critic_alpha(self):
        return torch.tensor([self.engine.cql_alpha_state()[3]])

    def act(self, obs, step, eval_mode):
        return self._act(np.asarray(obs, np.float32), step, eval_mode)

    def _noise(self, rows=None):
        raise RuntimeError('CQL draws five noise tensors; see _run_update')

    def _run_update(self, stddev):
        B, A, n = self.engine.batch, self.action_dim, self.n_samples
        if self.noise_hook is None:
            nc = na = None
        else:       # draw order cql.py:159,170-176,238; the hook returns N(0,1) except for the uniform(-1,1) random actions
            z_next = self.noise_hook((B, A))
            u_rand = self.noise_hook((n, B, A), 'uniform')
            z_cur, z_nxt = self.noise_hook((n, B, A)), self.noise_hook((n, B, A))
            nc = np.concatenate([np.asarray(x, np.float32).reshape(-1) for x in (z_next, u_rand, z_cur, z_nxt)])
            na = np.asarray(self.noise_hook((B, A)), np.float32)
        eng = self.engine
        if self.world_size == 1 or eng.comm is not None:
            eng.update(1.0, nc, na)
            return
        dist = torch.distributed
        if self.use_critic_lagrange:              # the multiplier steps on the GLOBAL penalty before any critic gradient is formed (cql.py:199-213)
            eng.update_phase(4, 1.0, nc, na)
            dist.all_reduce(eng.stats())          # this rank's sum of logsumexp and of Q1 + Q2
            eng.update_phase(5, 1.0, nc, na)
        else:
            eng.update_phase(0, 1.0, nc, na)
        dist.all_reduce(eng.flat(L.NET_CRITIC, L.T_GRAD))
        eng.update_phase(1, 1.0, nc, na)
        dist.all_reduce(eng.stats())              # sum of log_pi for the entropy temperature (cql.py:242-243)
        eng.update_phase(2, 1.0, nc, na)
        dist.all_reduce(eng.flat(L.NET_ACTOR, L.T_GRAD))
        eng.update_phase(3, 1.0, nc, na)

    def update(self, replay_iter, step):
        metrics = dict()
        self._step(replay_iter, 1.0)
        if self.use_tb:
            raw = self.engine.metrics_raw()
            if self.world_size > 1:
                t = torch.from_numpy(raw.copy()).to(self.engine.device)
                torch.distributed.all_reduce(t)
                raw = t.cpu().numpy()
            for idx, name in _CRITIC_METRICS + [(L.M_CRITIC_CQL, 'critic_cql'), (L.M_CRITIC_CQL_LOGSUM, 'critic_cql_logsum'),
                                                (L.M_ACTOR_ENT, 'actor_ent'), (L.M_ACTOR_ALPHA, 'actor_alpha'),
                                                (L.M_ACTOR_ALPHA_LOSS, 'actor_alpha_loss')]:
                metrics[name] = float(raw[idx])
        return metrics


class BCAgent(_AgentBase):
    KIND = 'bc'

    def __init__(self, name, obs_shape, action_shape, device, lr, hidden_dim, batch_size, stddev_schedule, use_tb,
                 has_next_action=False, *, precision='fp32', seed=0):
        self.lr = lr
        self.action_dim = action_shape[0]
        self.hidden_dim = hidden_dim
        self.device = device
        self.stddev_schedule = stddev_schedule
        self.use_tb = use_tb
        self._build(obs_shape[0], action_shape[0], hidden_dim, batch_size, lr, 0.0, 0.0, 0.0, device, precision, seed)
        self.train()

    def act(self, obs, step, eval_mode):
        return self._act(np.asarray(obs, np.float32), step, eval_mode)

    def update(self, replay_iter, step):
        metrics = dict()
        stddev = self._stddev(step)
        self._step(replay_iter, stddev)
        if self.use_tb:
            metrics.update(self._metrics([(L.M_BATCH_REWARD, 'batch_reward'), (L.M_ACTOR_LOSS, 'actor_loss')], stddev))
        return metrics


class DDPGAgent(_AgentBase):
    KIND = 'ddpg'

    def __init__(self, name, reward_free, obs_type, obs_shape, action_shape, device, lr, feature_dim, hidden_dim,
                 critic_target_tau, num_expl_steps, update_every_steps, stddev_schedule, nstep, batch_size, stddev_clip,
                 init_critic, use_tb, use_wandb, meta_dim=0, skill_type='uniform', *, precision='fp32', seed=0):
        if obs_type == 'pixels':
            if not (type(self) is DDPGAgent or getattr(self, '_PIXELS_OK', False)):
                raise NotImplementedError(f"exorl_amd: obs_type='pixels' is not built for {type(self).__name__} yet (DDPG, Proto, ICM, ICM-APT, "
                                          "Disagreement, DIAYN, APS, SMM and RND are)")
            return self._init_pixels(reward_free, obs_shape, action_shape, device, lr, feature_dim, hidden_dim, critic_target_tau, num_expl_steps,
                                     update_every_steps, stddev_schedule, batch_size, stddev_clip, init_critic, use_tb, use_wandb, precision, seed,
                                     meta_dim)
        if obs_type != 'states':
            raise NotImplementedError(f"exorl_amd DDPGAgent: unknown obs_type {obs_type!r}")
        self.reward_free = reward_free
        self.obs_type = obs_type
        self.obs_shape = obs_shape
        self.action_dim = action_shape[0]
        self.hidden_dim = hidden_dim
        self.lr = lr
        self.device = device
        self.critic_target_tau = critic_target_tau
        self.update_every_steps = update_every_steps
        self.use_tb = use_tb
        self.use_wandb = use_wandb
        self.num_expl_steps = num_expl_steps
        self.stddev_schedule = stddev_schedule
        self.stddev_clip = stddev_clip
        self.init_critic = init_critic
        self.feature_dim = feature_dim
        self.solved_meta = None
        self.obs_dim = obs_shape[0] + meta_dim
        self.aug = self.encoder = _Identity()
        self.encoder_opt = None
        self._precision = precision
        self._build(self.obs_dim, action_shape[0], hidden_dim, batch_size, lr, critic_target_tau, 0.0, stddev_clip, device,
                    precision, seed, **self._engine_kw())
        self.train()
        self.critic_target.train()

    def _engine_kw(self):
        return {}

    # ---- obs_type == 'pixels' (ddpg.py:12-39 Encoder, :42-123 pixel Actor/Critic, :213-328) -----------------------------------------
    def _init_pixels(self, reward_free, obs_shape, action_shape, device, lr, feature_dim, hidden_dim, critic_target_tau, num_expl_steps,
                     update_every_steps, stddev_schedule, batch_size, stddev_clip, init_critic, use_tb, use_wandb, precision, seed, meta_dim=0):
        self.reward_free, self.obs_type, self.obs_shape = reward_free, 'pixels', tuple(obs_shape)
        self._precision, self._pix_meta_dim = precision, meta_dim
        self.action_dim, self.hidden_dim, self.feature_dim = action_shape[0], hidden_dim, feature_dim
        self.lr, self.device, self.critic_target_tau = lr, device, critic_target_tau
        self.update_every_steps, self.use_tb, self.use_wandb = update_every_steps, use_tb, use_wandb
        self.num_expl_steps, self.stddev_schedule, self.stddev_clip, self.init_critic = num_expl_steps, stddev_schedule, stddev_clip, init_critic
        self.solved_meta = None
        self.world_size = 1
        if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
            raise NotImplementedError('exorl_amd: the pixel path is single-GPU this round')
        c = obs_shape[0]
        sf_dim = self._engine_kw().get('sf_dim', 0)         # APS: CriticSF heads (aps.py:94-104)
        w = _pixel_init(c, obs_shape[1], self.action_dim, feature_dim, hidden_dim, meta_dim, sf_dim)
        self.engine = PixelEngine(obs_shape, self.action_dim, feature_dim, hidden_dim, batch_size, lr=lr, tau=critic_target_tau,
                                  stddev_clip=stddev_clip, precision=precision, seed=seed, device=device, meta_dim=meta_dim, sf_dim=sf_dim)
        self.obs_dim = w['repr_dim'] + meta_dim          # ddpg.py:176 obs_dim = encoder.repr_dim + meta_dim
        conv_shapes = [s for l in range(4) for s in ((32, c if l == 0 else 32, 3, 3), (32,))]
        self.encoder = _PixelNetView(self.engine, 0, _ENC_KEYS, conv_shapes)
        self.actor = _PixelNetView(self.engine, 1, _PIX_ACTOR_KEYS)
        self.critic = _PixelNetView(self.engine, 2, _PIX_CRITIC_KEYS)
        self.critic_target = _PixelNetView(self.engine, 3, _PIX_CRITIC_KEYS)
        for view, ts in ((self.encoder, w['encoder']), (self.actor, w['actor']), (self.critic, w['critic'])):
            for p, t in zip(view.parameters(), ts):
                p.copy_(t.reshape(p.shape))
        self.engine.sync_target()
        self.aug = _Identity()
        self.encoder_opt = True            # the reference's `if self.encoder_opt is not None` checks hold for pixels
        self.noise_hook = None             # tests: callable(shape) -> standard normals for the TruncatedNormal draws
        self.shift_hook = None             # tests: callable(batch) -> (batch, 2) RandomShiftsAug draws
        self._slots = None
        self.training = True

    def _update_pixels(self, replay_iter, step):
        eng = self.engine
        if hasattr(replay_iter, 'sample_into'):
            self._slots = self._slots or eng.batch_slots()
            replay_iter.sample_into(self._slots, eng.batch)
        else:
            obs, action, reward, discount, next_obs = next(replay_iter)[:5]
            eng.set_batch(obs, action, reward, discount, next_obs)
        stddev = self._stddev(step)
        B, A = eng.batch, self.action_dim
        so = self.shift_hook(B) if self.shift_hook else None
        sn = self.shift_hook(B) if self.shift_hook else None
        nc = self.noise_hook((B, A)) if self.noise_hook else None
        na = self.noise_hook((B, A)) if self.noise_hook else None
        eng.update(stddev, so, sn, nc, na)
        metrics = dict()
        if self.use_tb or self.use_wandb:
            raw = eng.metrics_raw()
            for idx, name in _CRITIC_METRICS + [(L.M_ACTOR_LOGPROB, 'actor_logprob')]:
                metrics[name] = float(raw[idx])
            metrics['actor_ent'] = float(np.float32(0.5 + 0.5 * np.log(2 * np.pi) + np.log(stddev)) * self.action_dim)
        return metrics

    def train(self, training=True):
        if getattr(self, 'obs_type', 'states') == 'pixels':
            self.training = training
            for n in (self.encoder, self.actor, self.critic):
                n.train(training)
            return
        super().train(training)
        self.encoder.train(training)

    def init_from(self, other):
        if self.obs_type == 'pixels':
            utils.hard_update_params(other.encoder, self.encoder)
        utils.hard_update_params(other.actor, self.actor)
        if self.init_critic:          # critic.trunk = first 4 tensors (ddpg.py:209-210)
            for p, t in zip(other.critic.parameters()[:4], self.critic.parameters()[:4]):
                t.copy_(p)
        self.params_changed()

    def get_meta_specs(self):
        return tuple()

    def init_meta(self):
        return OrderedDict()

    def update_meta(self, meta, global_step, time_step, finetune=False):
        return meta

    def act(self, obs, meta, step, eval_mode):
        if self.obs_type == 'pixels':
            stddev = self._stddev(step)
            noise = self.noise_hook((1, self.action_dim)) if (self.noise_hook and not eval_mode) else None
            mv = np.concatenate([np.asarray(v, np.float32).reshape(-1) for v in meta.values()]) if self._pix_meta_dim else None
            a = self.engine.act(np.ascontiguousarray(obs), stddev, eval_mode, noise, mv)
            if not eval_mode and step < self.num_expl_steps:
                a.uniform_(-1.0, 1.0)
            return a.cpu().numpy()
        parts = [np.asarray(obs, np.float32).reshape(-1)] + [np.asarray(v, np.float32).reshape(-1) for v in meta.values()]
        return self._act(np.concatenate(parts), step, eval_mode)

    def update(self, replay_iter, step):
        metrics = dict()
        if step % self.update_every_steps != 0:      # ddpg.py:302-303 — no batch is consumed
            return metrics
        if self.obs_type == 'pixels':
            return self._update_pixels(replay_iter, step)
        stddev = self._stddev(step)
        self._step(replay_iter, stddev)
        if self.use_tb or self.use_wandb:
            metrics.update(self._metrics(_CRITIC_METRICS + [(L.M_ACTOR_LOGPROB, 'actor_logprob')], stddev))
        return metrics


def _seq_init(spec):
    """Initial tensors of an nn.Sequential-style module: spec = [('lin', in, out) | ('ln', dim)]. Same RNG consumption as the
    reference constructors: all Linears default-initialised at construction, then apply(utils.weight_init) in order."""
    mods = [nn.Linear(s[1], s[2]) if s[0] == 'lin' else nn.LayerNorm(s[1]) for s in spec]
    out = []
    for m in mods:
        if isinstance(m, nn.Linear):
            nn.init.orthogonal_(m.weight.data)
            m.bias.data.fill_(0.0)
        out += [m.weight.data, m.bias.data]
    return out


_RND_KEYS = [f'{n}.{i}.{w}' for n in ('predictor', 'target') for i in (1, 3, 5) for w in ('weight', 'bias')]
_ICM_KEYS = [f'{n}.{i}.{w}' for n in ('forward_net', 'backward_net') for i in (0, 2) for w in ('weight', 'bias')]
_APT_KEYS = ['trunk.0.weight', 'trunk.0.bias', 'trunk.1.weight', 'trunk.1.bias'] + _ICM_KEYS


class _RndView(NetView):
    """agent.rnd: the parameters plus BatchNorm1d's buffers under the reference's state_dict keys (rnd.py:24-27)."""

    def _bn(self):
        O = self._engine.obs_dim
        bn = self._engine.bn
        return bn[:O], bn[O:2 * O], bn[2 * O:]

    def state_dict(self):
        mean, var, cnt = self._bn()
        sd = OrderedDict([('normalize_obs.running_mean', mean.clone()), ('normalize_obs.running_var', var.clone()),
                          ('normalize_obs.num_batches_tracked', cnt.clone().long().reshape(()))])
        sd.update(super().state_dict())
        return sd

    def load_state_dict(self, sd, strict=True):
        sd = dict(sd)
        mean, var, cnt = self._bn()
        for k, dst in (('normalize_obs.running_mean', mean), ('normalize_obs.running_var', var), ('normalize_obs.num_batches_tracked', cnt)):
            if k in sd:
                dst.copy_(torch.as_tensor(sd.pop(k)).to(dst.device, torch.float32).reshape(dst.shape))
            elif strict:
                raise KeyError(f'state_dict mismatch: missing {k}')
        super().load_state_dict(sd, strict)


class _RmsView:
    """Stands where the reference has a utils.RMS object (agent.intrinsic_reward_rms / agent.pbe.rms): .M, .S, .n."""

    def __init__(self, engine):
        self._engine = engine

    M = property(lambda self: torch.tensor([self._engine.rms_state()[0]]))
    S = property(lambda self: torch.tensor([self._engine.rms_state()[1]]))
    n = property(lambda self: self._engine.rms_state()[2])


class _IntrAgent(DDPGAgent):
    """Shared update() of the reward-free agents (rnd.py:110-159, icm.py:94-139, icm_apt.py:112-158): module step and
    intrinsic reward on the sampled batch (libexorl_hip: exorl_intr_update), then the DDPG update on that reward."""
    LOSS_KEY = None

    def _intr_step(self):
        s = self._slots = self._slots or self.engine.batch_slots()
        self.intr.update(s.obs, s.action, s.next_obs, s.reward, s.reward, True)

    def enable_graph(self, replay_iter, step=0):
        return False                     # the module step is launched eagerly in front of the DDPG chain

    # ---- data parallel (last row) ----------------------------------------------------------------------------------------
    # What these modules compute is batch-global: BatchNorm statistics and the running RMS of the rewards (rnd.py:24,47-60,
    # utils.py:257-276), kNN particle entropy over the batch (utils.py:279-319), Sinkhorn normalisation and the candidate queue
    # (proto.py:114-157), SMM's mean_j log p*(s_j), and each steps its own optimiser on a batch mean. Under torch.distributed every rank
    # therefore all-gathers the sampled rows (one collective of B x (2 obs + act + 1) floats, <= 0.3 MB per rank at the shipped widths)
    # and runs the module step on the GLOBAL batch: the kernels are deterministic, so the replicas of the module stay bit-identical
    # without a gradient exchange, every statistic is the single-process one, and each rank keeps its own rows of the reward. The
    # actor / critic step — the 1024-wide layers, 35 of the step's ~45 GFLOP — runs sharded like the offline agents' (_run_update).
    @property
    def _module_batch(self):
        return self.engine.batch * getattr(self, 'world_size', 1)

    def _dp_buffers(self):
        if getattr(self, '_dp', None) is None:
            eng, ws = self.engine, self.world_size
            B, W, A = eng.batch, self.obs_dim, self.action_dim
            dev = eng.device
            pack = torch.empty(B, 2 * W + A + 1, device=dev)
            gpack = torch.empty(ws * B, 2 * W + A + 1, device=dev)
            g = dict(obs=torch.empty(ws * B, W, device=dev), action=torch.empty(ws * B, A, device=dev),
                     next_obs=torch.empty(ws * B, W, device=dev), reward=torch.empty(ws * B, device=dev))
            O = W - getattr(self, '_meta_dim', 0)
            slots = L.BatchOut(g['obs'].data_ptr(), W, g['action'].data_ptr(), A, g['reward'].data_ptr(), None, g['next_obs'].data_ptr(), W,
                               g['obs'].data_ptr() + 4 * O, W)
            self._dp = (pack, gpack, g, slots)
        return self._dp

    def _intr_step_dp(self):
        dist = torch.distributed
        eng, ws, rank = self.engine, self.world_size, torch.distributed.get_rank()
        B, W, A = eng.batch, self.obs_dim, self.action_dim
        local = self._slots = self._slots or eng.batch_slots()
        pack, gpack, g, gslots = self._dp_buffers()
        view = lambda ptr, cols: eng._view(ptr, B * cols).view(B, cols)
        pack[:, :W].copy_(view(local.obs, W))
        pack[:, W:2 * W].copy_(view(local.next_obs, W))
        pack[:, 2 * W:2 * W + A].copy_(view(local.action, A))
        pack[:, 2 * W + A].copy_(eng._view(local.reward, B))
        dist.all_gather(list(gpack.split(B)), pack)              # rows in rank order = the single-process batch of the equivalence test
        g['obs'].copy_(gpack[:, :W])
        g['next_obs'].copy_(gpack[:, W:2 * W])
        g['action'].copy_(gpack[:, 2 * W:2 * W + A])
        g['reward'].copy_(gpack[:, 2 * W + A])
        self._slots = gslots
        try:
            self._intr_step()                                    # module optimiser step + rewards of all ws * B rows, identical on every rank
        finally:
            self._slots = local
        eng._view(local.reward, B).copy_(g['reward'][rank * B:(rank + 1) * B])

    # ---- obs_type == 'pixels' ---------------------------------------------------------------------------------------------------
    # Every one of these agents augments and encodes obs and next_obs ONCE (icm.py:97-99, icm_apt.py:113-118, disagreement.py:98-100,
    # diayn.py:137-138), steps its module AND the encoder on the module's loss (encoder_opt.step() inside update_<module>), takes the
    # intrinsic reward from the updated module on the encodings computed before that step, and hands the critic and the actor those same
    # encodings detached (`update_critic(obs.detach(), ...)`): after the module step nothing else moves the encoder, and encoder_opt's
    # second .step() inside update_critic finds no gradients (Adam skips parameters whose .grad is None, step counts included).
    _PIX_GRAD = 0                        # which encoding carries the graph into the module's loss: 0 obs, 1 next_obs

    def _pix_module(self, fo, fn, s):
        """Module step + intrinsic reward on the encodings (device pointers); d(loss)/d(encoding) lands in self._dobs."""
        self.intr.update(fo, s.action, fn, s.reward, s.reward, True, dobs_out=self._dobs.data_ptr())

    def _pix_alloc(self):
        """Called by the subclass constructors once self.intr exists."""
        if getattr(self, 'obs_type', 'states') == 'pixels':
            self._dobs = torch.empty(self.engine.batch, self.obs_dim - self._pix_meta_dim, dtype=torch.float32, device=self.engine.device)

    def _update_pixels(self, replay_iter, step):
        eng = self.engine
        M = self._pix_meta_dim
        s = self._slots = self._slots or eng.batch_slots()
        if hasattr(replay_iter, 'sample_into'):
            replay_iter.sample_into(s, eng.batch)
        else:
            b = next(replay_iter)
            eng.set_batch(*b[:5])
            if M:
                eng.meta_rows().copy_(torch.as_tensor(b[5]).to(eng.device, torch.float32).reshape(eng.batch, M))
        B, A = eng.batch, self.action_dim
        eng.augment(self.shift_hook(B) if self.shift_hook else None, self.shift_hook(B) if self.shift_hook else None)
        fo, fn = eng.encode(0), eng.encode(1)
        if self.reward_free:
            self._pix_module(fo, fn, s)
            eng.encoder_step(self._PIX_GRAD, self._dobs.data_ptr(), 0)
        stddev = self._stddev(step)
        eng.set_train_encoder(False)
        eng.update(stddev, None, None, self.noise_hook((B, A)) if self.noise_hook else None, self.noise_hook((B, A)) if self.noise_hook else None,
                   keep_encoded=True)
        metrics = dict()
        if self.use_tb or self.use_wandb:
            raw = eng.metrics_raw()
            for idx, name in _CRITIC_METRICS + [(L.M_ACTOR_LOGPROB, 'actor_logprob')]:
                metrics[name] = float(raw[idx])
            metrics['actor_ent'] = float(np.float32(0.5 + 0.5 * np.log(2 * np.pi) + np.log(stddev)) * self.action_dim)
            if self.reward_free:
                ri = self.intr.metrics_raw()
                metrics[self.LOSS_KEY] = float(ri[L.IM_LOSS])
                metrics['intr_reward'] = float(ri[L.IM_INTR_REWARD])
                metrics['extr_reward'] = float(ri[L.IM_EXTR_REWARD])
                if self.LOSS_KEY == 'diayn_loss':
                    metrics['diayn_acc'] = float(ri[L.IM_ACC])
                if self.LOSS_KEY == 'aps_loss':
                    metrics['intr_ent_reward'] = float(ri[L.IM_ENT_REWARD])
                    metrics['intr_sf_reward'] = float(ri[L.IM_SF_REWARD])
            else:
                metrics['extr_reward'] = metrics['batch_reward']
        return metrics

    def update(self, replay_iter, step):
        metrics = dict()
        if step % self.update_every_steps != 0:
            return metrics
        if self.obs_type == 'pixels':
            return self._update_pixels(replay_iter, step)
        stddev = self._stddev(step)
        self._load_batch(replay_iter)
        if self.reward_free:
            if self.world_size != 1:
                self._intr_step_dp()
            else:
                self._intr_step()
        self._run_update(stddev)
        if self.use_tb or self.use_wandb:
            metrics.update(self._metrics(_CRITIC_METRICS + [(L.M_ACTOR_LOGPROB, 'actor_logprob')], stddev))
            if self.reward_free:
                raw = self.intr.metrics_raw()
                metrics[self.LOSS_KEY] = float(raw[L.IM_LOSS])
                metrics['intr_reward'] = float(raw[L.IM_INTR_REWARD])
                metrics['extr_reward'] = float(raw[L.IM_EXTR_REWARD])
                if self.LOSS_KEY == 'diayn_loss':
                    metrics['diayn_acc'] = float(raw[L.IM_ACC])
                if self.LOSS_KEY == 'aps_loss':
                    metrics['intr_ent_reward'] = float(raw[L.IM_ENT_REWARD])
                    metrics['intr_sf_reward'] = float(raw[L.IM_SF_REWARD])
                if self.LOSS_KEY == 'rnd_loss':
                    metrics['pred_error_mean'] = float(raw[L.IM_RMS_MEAN])
                    metrics['pred_error_std'] = float(raw[L.IM_RMS_STD])
            else:
                metrics['extr_reward'] = metrics['batch_reward']
        return metrics


class _RndPixelView(_RndView):
    """agent.rnd on pixels (rnd.py:13-45): normalize_obs is a BatchNorm2d over the frames (buffers in the pixel engine), predictor.0 IS the
    agent's encoder, target.0 its frozen copy (the pixel engine's encoder_target slot); the six Linear layers live in the module engine."""

    def __init__(self, intr, pixel, encoder_view, target_view):
        super().__init__(intr, None, _RND_KEYS)
        self._pixel, self._enc, self._tgt = pixel, encoder_view, target_view

    def _bn(self):
        b = self._pixel.bn2d()
        c = (b.numel() - 1) // 2
        return b[:c], b[c:2 * c], b[2 * c:]

    def state_dict(self):
        sd = super().state_dict()
        for pre, view in (('predictor.0.', self._enc), ('target.0.', self._tgt)):
            sd.update({pre + k: v for k, v in view.state_dict().items()})
        return sd

    def load_state_dict(self, sd, strict=True):
        sd = dict(sd)
        for pre, view in (('predictor.0.', self._enc), ('target.0.', self._tgt)):
            sub = {k[len(pre):]: sd.pop(k) for k in list(sd) if k.startswith(pre)}
            if sub or strict:
                view.load_state_dict(sub, strict)
        super().load_state_dict(sd, strict)


class RNDAgent(_IntrAgent):
    """agents/unsupervised_learning/rnd.py:63-159 (configs/agent/rnd.yaml)."""
    LOSS_KEY = 'rnd_loss'
    _PIXELS_OK = True

    def __init__(self, rnd_rep_dim, update_encoder, rnd_scale=1., **kwargs):
        super().__init__(**kwargs)
        self.rnd_scale = rnd_scale
        self.update_encoder = update_encoder
        O, H = self.obs_dim, self.hidden_dim
        pixels = self.obs_type == 'pixels'
        if pixels:      # RND(...) construction order (rnd.py:28-45): the six Linears, then weight_init over predictor (the shared encoder's
            # convolutions are drawn AGAIN, then its Linears) and target (the copied encoder's convolutions get draws of their own)
            lins = [nn.Linear(*d) for d in ((O, H), (H, H), (H, rnd_rep_dim)) * 2]
            c = self.obs_shape[0]
            shapes = [(32, c if l == 0 else 32, 3, 3) for l in range(4)]
            w, convs = [], []
            for half in range(2):
                cw = [nn.init.orthogonal_(torch.empty(sh), nn.init.calculate_gain('relu')) for sh in shapes]
                convs.append(cw)
                for m in lins[3 * half:3 * half + 3]:
                    nn.init.orthogonal_(m.weight.data)
                    m.bias.data.fill_(0.0)
                    w += [m.weight.data, m.bias.data]
            for view_p, cw in zip(self.encoder.parameters()[0::2], convs[0]):
                view_p.copy_(cw.reshape(view_p.shape))
            for view_p in self.encoder.parameters()[1::2]:
                view_p.zero_()
        else:
            w = _seq_init([('lin', O, H), ('lin', H, H), ('lin', H, rnd_rep_dim)] * 2)      # predictor then target (rnd.py:28-43)
        self.intr = IntrEngine('rnd', O, self.action_dim, H, self._module_batch, rep_dim=rnd_rep_dim, lr=self.lr, scale=rnd_scale,
                               precision=self._precision, device=self.device, encoded=pixels)
        if pixels:
            conv_shapes = [s_ for l in range(4) for s_ in ((32, self.obs_shape[0] if l == 0 else 32, 3, 3), (32,))]
            self.rnd_target_encoder = _ParamList(_ENC_KEYS, self.engine.encoder_target_tensors(conv_shapes))
            for p, cw in zip(self.rnd_target_encoder.parameters()[0::2], convs[1]):
                p.copy_(cw)
            for p in self.rnd_target_encoder.parameters()[1::2]:
                p.zero_()
            self.rnd = _RndPixelView(self.intr, self.engine, self.encoder, self.rnd_target_encoder)
            NetView.load_state_dict(self.rnd, {k: t for k, t in zip(_RND_KEYS, w)})
        else:
            self.rnd = _RndView(self.intr, None, _RND_KEYS)
            for p, t in zip(self.rnd.parameters(), w):
                p.copy_(t.reshape(p.shape))
        self.intrinsic_reward_rms = _RmsView(self.intr)
        self._pix_alloc()

    def _update_pixels(self, replay_iter, step):
        """rnd.py:110-159 on pixels. RND.forward augments the raw frames itself, normalises them with a BatchNorm2d and runs the agent's
        encoder inside its predictor (and a frozen copy inside its target): update_rnd steps that encoder twice on the same gradients
        (rnd_opt, then encoder_opt), compute_intr_reward draws another augmentation and runs the moved encoder, and only then are obs and
        next_obs augmented and encoded for the critic and the actor (detached)."""
        eng = self.engine
        s = self._slots = self._slots or eng.batch_slots()
        if hasattr(replay_iter, 'sample_into'):
            replay_iter.sample_into(s, eng.batch)
        else:
            eng.set_batch(*next(replay_iter)[:5])
        B, A = eng.batch, self.action_dim
        sh = lambda: self.shift_hook(B) if self.shift_hook else None
        if self.reward_free:
            fp, ft = eng.rnd_features(sh(), 5.0)
            self.intr.update(fp, None, ft, s.reward, s.reward, 2, dobs_out=self._dobs.data_ptr())
            eng.encoder_step(0, self._dobs.data_ptr(), 2)
            fp, ft = eng.rnd_features(sh(), 5.0)
            self.intr.update(fp, None, ft, s.reward, s.reward, False)
        stddev = self._stddev(step)
        eng.set_train_encoder(False)
        eng.update(stddev, sh(), sh(), self.noise_hook((B, A)) if self.noise_hook else None, self.noise_hook((B, A)) if self.noise_hook else None)
        metrics = dict()
        if self.use_tb or self.use_wandb:
            raw = eng.metrics_raw()
            for idx, name in _CRITIC_METRICS + [(L.M_ACTOR_LOGPROB, 'actor_logprob')]:
                metrics[name] = float(raw[idx])
            metrics['actor_ent'] = float(np.float32(0.5 + 0.5 * np.log(2 * np.pi) + np.log(stddev)) * self.action_dim)
            ri = self.intr.metrics_raw()
            if self.reward_free:
                metrics['rnd_loss'] = float(ri[L.IM_LOSS])
                metrics['intr_reward'] = float(ri[L.IM_INTR_REWARD])
                metrics['extr_reward'] = float(ri[L.IM_EXTR_REWARD])
            else:
                metrics['extr_reward'] = metrics['batch_reward']
            M, S_, _n = self.intr.rms_state()
            metrics['pred_error_mean'] = float(M)
            metrics['pred_error_std'] = float(np.sqrt(np.float32(S_)))
        return metrics


class ICMAgent(_IntrAgent):
    """agents/unsupervised_learning/icm.py:48-139 (configs/agent/icm.yaml)."""
    LOSS_KEY = 'icm_loss'
    _PIXELS_OK = True

    def __init__(self, icm_scale, update_encoder, **kwargs):
        super().__init__(**kwargs)
        self.icm_scale = icm_scale
        self.update_encoder = update_encoder
        O, A, H = self.obs_dim, self.action_dim, self.hidden_dim
        w = _seq_init([('lin', O + A, H), ('lin', H, O), ('lin', 2 * O, H), ('lin', H, A)])
        self.intr = IntrEngine('icm', O, A, H, self._module_batch, lr=self.lr, scale=icm_scale, precision=self._precision,
                               device=self.device)
        self.icm = NetView(self.intr, None, _ICM_KEYS)
        for p, t in zip(self.icm.parameters(), w):
            p.copy_(t.reshape(p.shape))
        self._pix_alloc()


class _PbeView:
    def __init__(self, engine):
        self.rms = _RmsView(engine)


class ICMAPTAgent(_IntrAgent):
    """agents/unsupervised_learning/icm_apt.py:60-158 (configs/agent/icm_apt.yaml)."""
    LOSS_KEY = 'icm_loss'
    _PIXELS_OK = True

    def __init__(self, icm_scale, knn_rms, knn_k, knn_avg, knn_clip, update_encoder, icm_rep_dim, **kwargs):
        super().__init__(**kwargs)
        self.icm_scale = icm_scale
        self.update_encoder = update_encoder
        O, A, H, R = self.obs_dim, self.action_dim, self.hidden_dim, icm_rep_dim
        w = _seq_init([('lin', O, R), ('ln', R), ('lin', R + A, H), ('lin', H, R), ('lin', 2 * R, H), ('lin', H, A)])
        self.intr = IntrEngine('icm_apt', O, A, H, self._module_batch, rep_dim=R, lr=self.lr, scale=icm_scale, knn_k=knn_k,
                               knn_avg=knn_avg, knn_rms=knn_rms, knn_clip=knn_clip, precision=self._precision, device=self.device)
        self.icm = NetView(self.intr, None, _APT_KEYS)
        for p, t in zip(self.icm.parameters(), w):
            p.copy_(t.reshape(p.shape))
        self.pbe = _PbeView(self.intr)
        self._pix_alloc()


_DIS_KEYS = [f'ensemble.{m}.{i}.{w}' for m in range(5) for i in (0, 2) for w in ('weight', 'bias')]
_DIAYN_KEYS = [f'skill_pred_net.{i}.{w}' for i in (0, 2, 4) for w in ('weight', 'bias')]


class DisagreementAgent(_IntrAgent):
    """agents/unsupervised_learning/disagreement.py:50-136 (configs/agent/disagreement.yaml)."""
    LOSS_KEY = 'disagreement_loss'
    _PIXELS_OK = True

    def __init__(self, update_encoder, **kwargs):
        super().__init__(**kwargs)
        self.update_encoder = update_encoder
        O, A, H = self.obs_dim, self.action_dim, self.hidden_dim
        # the ensemble keeps nn.Linear's default initialisation: Disagreement never applies utils.weight_init (disagreement.py:12-18)
        w = []
        for _ in range(5):
            for m in (nn.Linear(O + A, H), nn.Linear(H, O)):
                w += [m.weight.data, m.bias.data]
        self.intr = IntrEngine('disagreement', O, A, H, self._module_batch, lr=self.lr, n_models=5, precision=self._precision,
                               device=self.device)
        self.disagreement = NetView(self.intr, None, _DIS_KEYS)
        for p, t in zip(self.disagreement.parameters(), w):
            p.copy_(t.reshape(p.shape))
        self._pix_alloc()


class _Spec:
    """Stand-in for dm_env.specs.Array as the replay storage reads it (name, shape, dtype)."""

    def __init__(self, shape, dtype, name):
        self.shape, self.dtype, self.name = tuple(shape), np.dtype(dtype), name


class _MetaObsMixin:
    """Agents whose batch carries a 6th tensor (DIAYN's skill, APS's task) that is appended to obs / next_obs for the actor and
    critic (diayn.py:162-164, aps.py:236-238) while the module sees the raw observation: the agent's batch slots hold
    [obs | meta] rows, the module gets strided views of them."""
    _meta_dim = 0

    def _views(self):
        s = self._slots = self._slots or self.engine.batch_slots()
        B, W = self.engine.batch, self.obs_dim
        return s, self.engine._view(s.obs, B * W).view(B, W), self.engine._view(s.next_obs, B * W).view(B, W)

    def _load_batch(self, replay_iter):
        O, S = self.obs_dim - self._meta_dim, self._meta_dim
        s, obs_v, next_v = self._views()
        if hasattr(replay_iter, 'sample_into'):      # HBM sampler: obs and the meta columns land in the [obs | meta] rows
            out = L.BatchOut(s.obs, s.obs_stride, s.action, s.action_stride, s.reward, s.discount, s.next_obs, s.next_obs_stride,
                             s.obs + 4 * O, O + S)
            replay_iter.sample_into(out, self.engine.batch)
            next_v[:, O:].copy_(obs_v[:, O:])
            return
        obs, action, reward, discount, next_obs, meta = [torch.as_tensor(x).to(self.engine.device, torch.float32) for x in next(replay_iter)[:6]]
        self.engine.set_batch(torch.cat([obs, meta], 1), action, reward, discount, torch.cat([next_obs, meta], 1))

    def _intr_step(self):
        O, W = self.obs_dim - self._meta_dim, self.obs_dim
        s = self._slots
        self.intr.update(s.obs, None, s.next_obs, s.reward, s.reward, True, skill=s.obs + 4 * O, obs_ld=W, next_obs_ld=W, skill_ld=W)


class DIAYNAgent(_MetaObsMixin, _IntrAgent):
    """agents/unsupervised_learning/diayn.py:32-176 (configs/agent/diayn.yaml): the skill rides in the batch as a 6th tensor and
    is appended to obs / next_obs for the actor and critic; the discriminator sees the raw next_obs."""
    LOSS_KEY = 'diayn_loss'
    _PIXELS_OK = True
    _PIX_GRAD = 1

    def __init__(self, update_skill_every_step, skill_dim, diayn_scale, update_encoder, **kwargs):
        self.skill_dim = self._meta_dim = skill_dim
        self.update_skill_every_step = update_skill_every_step
        self.diayn_scale = diayn_scale
        self.update_encoder = update_encoder
        kwargs['meta_dim'] = self.skill_dim
        self.skill_type = kwargs['skill_type']
        super().__init__(**kwargs)
        O, H = self.obs_dim - self.skill_dim, self.hidden_dim
        w = _seq_init([('lin', O, H), ('lin', H, H), ('lin', H, skill_dim)])
        self.intr = IntrEngine('diayn', O, self.action_dim, H, self._module_batch, rep_dim=skill_dim, lr=self.lr, scale=diayn_scale,
                               precision=self._precision, device=self.device)
        self.diayn = NetView(self.intr, None, _DIAYN_KEYS)
        for p, t in zip(self.diayn.parameters(), w):
            p.copy_(t.reshape(p.shape))
        self._pix_alloc()

    def _pix_module(self, fo, fn, s):          # the discriminator reads the next frame's encoding (diayn.py:141-147)
        self.intr.update(fo, None, fn, s.reward, s.reward, True, skill=s.meta, skill_ld=self.skill_dim, dobs_out=self._dobs.data_ptr())

    def get_meta_specs(self):
        return (_Spec((self.skill_dim,), np.float32, 'skill'),)

    def init_meta(self):
        if self.skill_type == 'uniform':
            skill = np.random.uniform(0, 1, self.skill_dim).astype(np.float32)
        else:
            skill = np.zeros(self.skill_dim, dtype=np.float32)
            skill[np.random.choice(self.skill_dim)] = 1.0
        meta = OrderedDict()
        meta['skill'] = skill
        return meta

    def update_meta(self, meta, global_step, time_step, finetune=False):
        if global_step % self.update_skill_every_step == 0:
            return self.init_meta()
        return meta


_APS_KEYS = [f'state_feat_net.{i}.{w}' for i in (0, 2, 4) for w in ('weight', 'bias')]


class APSAgent(_MetaObsMixin, _IntrAgent):
    """agents/unsupervised_learning/aps.py:82-320 (configs/agent/aps.yaml): DDPG whose critic emits sf_dim successor features per
    head, Q = task . features (CriticSF, aps.py:12-60); the task vector rides in the batch and in the trailing columns of obs."""
    KIND = 'aps'
    LOSS_KEY = 'aps_loss'
    _PIXELS_OK = True
    _PIX_GRAD = 1                        # update_aps reaches the encoder through next_obs (aps.py:147-159,203-204)

    def _pix_module(self, fo, fn, s):
        self.intr.update(fo, None, fn, s.reward, s.reward, True, skill=s.meta, skill_ld=self.sf_dim, dobs_out=self._dobs.data_ptr())

    def __init__(self, update_task_every_step, sf_dim, knn_rms, knn_k, knn_avg, knn_clip, num_init_steps, lstsq_batch_size, update_encoder,
                 **kwargs):
        self.sf_dim = self._meta_dim = sf_dim
        self.update_task_every_step = update_task_every_step
        self.num_init_steps = num_init_steps
        self.lstsq_batch_size = lstsq_batch_size
        self.update_encoder = update_encoder
        kwargs['meta_dim'] = self.sf_dim
        super().__init__(**kwargs)
        O, H = self.obs_dim - self.sf_dim, self.hidden_dim
        w = _seq_init([('lin', O, H), ('lin', H, H), ('lin', H, sf_dim)])
        self.intr = IntrEngine('aps', O, self.action_dim, H, self._module_batch, rep_dim=sf_dim, lr=self.lr, knn_k=knn_k, knn_avg=knn_avg,
                               knn_rms=knn_rms, knn_clip=knn_clip, precision=self._precision, device=self.device)
        self.aps = NetView(self.intr, None, _APS_KEYS)
        for p, t in zip(self.aps.parameters(), w):
            p.copy_(t.reshape(p.shape))
        self.pbe = _PbeView(self.intr)
        self._pix_alloc()

    def _engine_kw(self):
        return {'sf_dim': self.sf_dim}

    def get_meta_specs(self):
        return (_Spec((self.sf_dim,), np.float32, 'task'),)

    def init_meta(self):
        if self.solved_meta is not None:
            return self.solved_meta
        task = torch.randn(self.sf_dim)
        task = task / torch.norm(task)
        meta = OrderedDict()
        meta['task'] = task.cpu().numpy()
        return meta

    def update_meta(self, meta, global_step, time_step, finetune=False):
        if global_step % self.update_task_every_step == 0:
            return self.init_meta()
        return meta

    @torch.no_grad()
    def regress_meta(self, replay_iter, step):
        """aps.py:247-266: least-squares task from (reward, phi(obs)) pairs at the start of fine-tuning. Host-side, once per run;
        the features come from the module's forward pass (reward-only call on zero tasks), the solve is torch.linalg.lstsq."""
        obs, reward = [], []
        n = 0
        while n < self.lstsq_batch_size:
            batch = next(replay_iter)
            r = torch.as_tensor(batch[2]).to(self.device, torch.float32)
            if self.obs_type == 'pixels':          # aug_and_encode (aps.py:262) through the engine, one batch of frames at a time
                eng = self.engine
                eng.set_batch(*batch[:5])
                eng.augment(self.shift_hook(eng.batch) if self.shift_hook else None, self.shift_hook(eng.batch) if self.shift_hook else None)
                o = eng.feature_view(eng.encode(0)).clone()
            else:
                o = torch.as_tensor(batch[0]).to(self.device, torch.float32)
            obs.append(o)
            reward.append(r.reshape(-1, 1))
            n += o.shape[0]
        obs, reward = torch.cat(obs, 0), torch.cat(reward, 0)
        rep = self._features(obs)
        task = torch.linalg.lstsq(reward, rep)[0][:rep.size(1), :][0]
        task = task / torch.norm(task)
        meta = OrderedDict()
        meta['task'] = task.cpu().numpy()
        self.solved_meta = meta
        return meta

    def _features(self, obs):
        """F.normalize(state_feat_net(obs)) for arbitrary row counts: three Linear layers on the module's own parameter views
        (inference outside the update path; torch GEMMs are plumbing here, not the hot path)."""
        p = self.aps.parameters()
        h = torch.relu(obs @ p[0].t() + p[1])
        h = torch.relu(h @ p[2].t() + p[3])
        return torch.nn.functional.normalize(h @ p[4].t() + p[5], dim=-1)


_SMM_KEYS = ([f'z_pred_net.{i}.{w}' for i in (0, 2, 4) for w in ('weight', 'bias')] +
             [f'vae.{n}.{w}' for n in ('enc.0', 'enc.2', 'enc_mu', 'enc_logvar', 'dec.0', 'dec.2', 'dec.4') for w in ('weight', 'bias')])


def _smm_init(O, Z, H, code_dim=128, vae_hidden=150):
    """SMM.__init__ (smm.py:89-104): z_pred_net, then VAE (which applies weight_init to itself), then weight_init over everything —
    the VAE's Linear layers are re-drawn a second time."""
    def orth(ms):
        for m in ms:
            nn.init.orthogonal_(m.weight.data)
            m.bias.data.fill_(0.0)
    W = O + Z
    zp = [nn.Linear(O, H), nn.Linear(H, H), nn.Linear(H, Z)]
    vae = [nn.Linear(W, vae_hidden), nn.Linear(vae_hidden, vae_hidden), nn.Linear(vae_hidden, code_dim), nn.Linear(vae_hidden, code_dim),
           nn.Linear(code_dim, vae_hidden), nn.Linear(vae_hidden, vae_hidden), nn.Linear(vae_hidden, W)]
    orth(vae)
    orth(zp)
    orth(vae)
    return [t for m in zp + vae for t in (m.weight.data, m.bias.data)]


class SMMAgent(_MetaObsMixin, _IntrAgent):
    """agents/unsupervised_learning/smm.py:115-281 (configs/agent/smm.yaml) on state observations.

    Bug-compatible with the reference in one documented place: smm.py adds the 1-D `log_p_star` (B,) to (B,1) terms, so its reward and
    TD target are (B,B) matrices and `mse_loss` averages over all pairs. Per sample that is the TD loss against
    rest_i + mean_j log_p_star_j, plus var_j(log_p_star_j) in each critic's loss value; the module writes exactly that reward and
    `critic_loss` carries the variance term (tests/golden/tiny_smm.npz reproduces the reference's numbers)."""
    LOSS_KEY = 'loss_vae'
    _PIXELS_OK = True                    # smm.py:264-331 with obs_type == 'pixels': the VAE and the skill predictor read the encoding, p*(s) is dropped
    _PIX_GRAD = 0

    def __init__(self, z_dim, sp_lr, vae_lr, vae_beta, state_ent_coef, latent_ent_coef, latent_cond_ent_coef, update_encoder, **kwargs):
        self.z_dim = self._meta_dim = z_dim
        self.state_ent_coef = state_ent_coef
        self.latent_ent_coef = latent_ent_coef
        self.latent_cond_ent_coef = latent_cond_ent_coef
        self.update_encoder = update_encoder
        kwargs['meta_dim'] = self.z_dim
        super().__init__(**kwargs)
        O, H = self.obs_dim - z_dim, self.hidden_dim
        self.goal = (150, 75)
        w = _smm_init(O, z_dim, H)
        self.intr = IntrEngine('smm', O, self.action_dim, H, self._module_batch, rep_dim=z_dim, sp_lr=sp_lr, vae_lr=vae_lr, vae_beta=vae_beta,
                               state_ent_coef=state_ent_coef, latent_ent_coef=latent_ent_coef, latent_cond_ent_coef=latent_cond_ent_coef,
                               goal=self.goal, precision=self._precision, device=self.device, encoded=self.obs_type == 'pixels')
        self.smm = NetView(self.intr, None, _SMM_KEYS)
        for p, t in zip(self.smm.parameters(), w):
            p.copy_(t.reshape(p.shape))
        self.ft_returns = np.zeros(z_dim, dtype=np.float32)
        self.ft_not_finished = [True for _ in range(z_dim)]
        self.eps_hook = None            # tests: callable(shape) -> the VAE's epsilon (torch.randn in smm.py:62)
        self._pix_alloc()

    def _eps(self):
        if self.eps_hook is None:
            return None
        return torch.as_tensor(np.asarray(self.eps_hook((self.engine.batch, 128)), np.float32), device=self.engine.device).contiguous()

    def _pix_module(self, fo, fn, s):
        """update_vae on obs_z = [encoding | z] (the encoder steps on the VAE's loss, smm.py:173-185), update_pred on the detached encoding."""
        eng = self.engine
        xz = torch.cat([eng.feature_view(fo), eng.meta_rows()], 1).contiguous()
        e = self._eps()
        self.intr.update(xz.data_ptr(), None, None, s.reward, s.reward, True, skill=s.meta, obs_ld=xz.shape[1], skill_ld=self.z_dim,
                         cat_uniform=e.data_ptr() if e is not None else None, dobs_out=self._dobs.data_ptr())
        self._keep_eps = (e, xz)

    def get_meta_specs(self):
        return (_Spec((self.z_dim,), np.float32, 'z'),)

    def init_meta(self):
        z = np.zeros(self.z_dim, dtype=np.float32)
        z[np.random.choice(self.z_dim)] = 1.0
        meta = OrderedDict()
        meta['z'] = z
        return meta

    def update_meta(self, meta, global_step, time_step, finetune=False):
        if self.reward_free:                             # smm.py:154-160 (sic: the fine-tuning rule is keyed on reward_free)
            return self.update_meta_ft(meta, global_step, time_step)
        if time_step.last():
            return self.init_meta()
        return meta

    def update_meta_ft(self, meta, global_step, time_step):
        z_ind = meta['z'].argmax()
        if any(self.ft_not_finished):
            self.ft_returns[z_ind] += time_step.reward
            if time_step.last():
                if not any(self.ft_not_finished):
                    new_z_ind = self.ft_returns.argmax()
                else:
                    self.ft_not_finished[z_ind] = False
                    not_tried_z = sum(self.ft_not_finished)
                    for i in range(self.z_dim):
                        if self.ft_not_finished[i]:
                            if np.random.random() < 1 / not_tried_z:
                                new_z_ind = i
                                break
                            not_tried_z -= 1
                new_z = np.zeros(self.z_dim, dtype=np.float32)
                new_z[new_z_ind] = 1.0
                meta['z'] = new_z
        return meta

    def _intr_step(self):
        O, W = self.obs_dim - self._meta_dim, self.obs_dim
        s = self._slots
        e = None
        if self.eps_hook is not None:
            e = torch.as_tensor(np.asarray(self.eps_hook((self.intr.batch, 128)), np.float32), device=self.engine.device).contiguous()
        self.intr.update(s.obs, None, None, s.reward, s.reward, True, skill=s.obs + 4 * O, obs_ld=W, skill_ld=W,
                         cat_uniform=e.data_ptr() if e is not None else None)
        self._keep_eps = e

    def update(self, replay_iter, step):
        if step % self.update_every_steps != 0:
            return dict()
        metrics = super().update(replay_iter, step)
        if self.obs_type == 'pixels':                    # smm.py:260-265: loss_vae / loss_pred ride with use_tb, nothing else is added
            if self.reward_free and (self.use_tb or self.use_wandb):
                metrics['loss_pred'] = float(self.intr.metrics_raw()[5])
            return metrics
        if self.reward_free:                             # smm.py:249-258: these are reported whatever use_tb says
            raw = self.intr.metrics_raw()
            for k in ('icm_loss', 'loss_vae'):
                metrics.pop(k, None)
            metrics.update(intr_reward=float(raw[1]), log_p_star=float(raw[3]), pred_log_ratios=float(raw[4]),
                           latent_ent_coef=float(np.float32(self.latent_ent_coef) * np.float32(np.log(self.z_dim))),
                           latent_cond_ent_coef=float(raw[6]), loss_vae=float(raw[0]), loss_pred=float(raw[5]))
            if 'critic_loss' in metrics:
                metrics['critic_loss'] += 2.0 * float(raw[7])
        return metrics


class _TensorsView(NetView):
    """A NetView over a subset of an engine's tensors (agent.predictor / .projector / .protos / .predictor_target)."""

    def __init__(self, engine, indices, keys):
        self._engine, self._net, self._keys, self._on_change = engine, None, list(keys), None
        self.training = True
        self._idx = list(indices)
        self._params = [engine.tensor(None, i, L.T_PARAM) for i in self._idx]

    def grads(self):
        return [self._engine.tensor(None, i, L.T_GRAD) for i in self._idx]


def _proto_init(O, pred_dim, proj_dim, num_protos):
    """proto.py:55-67: predictor (Linear + weight_init), projector (Projector applies weight_init itself, then the agent applies it
    again), protos (bias-free Linear + weight_init) — the same RNG consumption, tensor by tensor."""
    def orth(m):
        nn.init.orthogonal_(m.weight.data)
        if m.bias is not None:
            m.bias.data.fill_(0.0)
    pred = nn.Linear(O, pred_dim)
    orth(pred)
    p0, p2 = nn.Linear(pred_dim, proj_dim), nn.Linear(proj_dim, pred_dim)
    for _ in range(2):
        orth(p0)
        orth(p2)
    protos = nn.Linear(pred_dim, num_protos, bias=False)
    orth(protos)
    return [pred.weight.data, pred.bias.data, p0.weight.data, p0.bias.data, p2.weight.data, p2.bias.data, protos.weight.data]


class ProtoAgent(_IntrAgent):
    """agents/unsupervised_learning/proto.py:46-207 (configs/agent/proto.yaml) on state observations: the encoder is the identity,
    so encoder_target and the encoder's share of proto_opt vanish; the intrinsic reward is computed on next_obs (proto.py:175-177)."""
    LOSS_KEY = 'repr_loss'
    _PIXELS_OK = True

    def __init__(self, pred_dim, proj_dim, queue_size, num_protos, tau, encoder_target_tau, topk, update_encoder, **kwargs):
        super().__init__(**kwargs)
        self.tau = tau
        self.encoder_target_tau = encoder_target_tau
        self.topk = topk
        self.num_protos = num_protos
        self.update_encoder = update_encoder
        self.encoder_target = _Identity()
        self._precision = kwargs.get('precision', 'fp32')
        if self.obs_type == 'pixels':            # encoder_target = deepcopy(encoder) (proto.py:55): no RNG draws
            c = self.obs_shape[0]
            shapes = [s for l in range(4) for s in ((32, c if l == 0 else 32, 3, 3), (32,))]
            self.engine.encoder_target(init=True)
            self.encoder_target = _ParamList(_ENC_KEYS, self.engine.encoder_target_tensors(shapes))
            self._dobs = torch.zeros(self.engine.batch, self.obs_dim, device=self.engine.device)
        O = self.obs_dim
        w = _proto_init(O, pred_dim, proj_dim, num_protos)
        self.intr = IntrEngine('proto', O, self.action_dim, proj_dim, self._module_batch, rep_dim=pred_dim, lr=self.lr, knn_k=topk,
                               num_protos=num_protos, queue_size=queue_size, tau=tau, target_tau=encoder_target_tau,
                               precision=self._precision, device=self.device)
        self.predictor = _TensorsView(self.intr, [0, 1], ['weight', 'bias'])
        self.projector = _TensorsView(self.intr, [2, 3, 4, 5], ['trunk.0.weight', 'trunk.0.bias', 'trunk.2.weight', 'trunk.2.bias'])
        self.protos = _TensorsView(self.intr, [6], ['weight'])
        self.predictor_target = _TensorsView(self.intr, [7, 8], ['weight', 'bias'])
        for view, ts in ((self.predictor, w[0:2]), (self.projector, w[2:6]), (self.protos, w[6:7]), (self.predictor_target, w[0:2])):
            for p, t in zip(view.parameters(), ts):
                p.copy_(t.reshape(p.shape))
        self.queue = self.intr.queue
        self.cat_hook = None            # tests: callable(num_protos) -> uniforms standing in for Categorical(prob).sample()

    queue_ptr = property(lambda self: self.intr.queue_ptr(), lambda self, v: self.intr.queue_ptr(v))

    def init_from(self, other):         # proto.py:87-96
        if self.obs_type == 'pixels':
            utils.hard_update_params(other.encoder, self.encoder)
        utils.hard_update_params(other.actor, self.actor)
        utils.hard_update_params(other.predictor, self.predictor)
        utils.hard_update_params(other.projector, self.projector)
        utils.hard_update_params(other.protos, self.protos)
        if self.init_critic:
            utils.hard_update_params(other.critic, self.critic)
        if self.obs_type != 'pixels':
            self.params_changed()

    def _cat_u(self):
        if self.cat_hook is None:
            return None
        return torch.as_tensor(np.asarray(self.cat_hook(self.num_protos), np.float32), device=self.engine.device)

    def _intr_step(self):
        s = self._slots = self._slots or self.engine.batch_slots()
        u = self._cat_u()
        self.intr.update(s.obs, None, s.next_obs, s.reward, s.reward, True, cat_uniform=u.data_ptr() if u is not None else None)
        self._keep_u = u

    def _update_pixels(self, replay_iter, step):
        """proto.py:159-207 on pixels: augment once; the proto step reaches the encoder through proto_opt; reward from the re-encoded
        next_obs; DDPG step with the encoding detached in update_critic; Polyak updates."""
        eng = self.engine
        if hasattr(replay_iter, 'sample_into'):
            self._slots = self._slots or eng.batch_slots()
            replay_iter.sample_into(self._slots, eng.batch)
        else:
            obs, action, reward, discount, next_obs = next(replay_iter)[:5]
            eng.set_batch(obs, action, reward, discount, next_obs)
        s = self._slots = self._slots or eng.batch_slots()
        B, A = eng.batch, self.action_dim
        eng.augment(self.shift_hook(B) if self.shift_hook else None, self.shift_hook(B) if self.shift_hook else None)
        if self.reward_free:
            fo = eng.encode(0)
            ft = eng.encode(1, target=True)
            self.intr.update(fo, None, ft, None, s.reward, 2, next_obs_target=ft, dobs_out=self._dobs.data_ptr())
            eng.encoder_step(0, self._dobs.data_ptr(), 1)
            fn = eng.encode(1)
            u = self._cat_u()
            self.intr.update(fo, None, fn, s.reward, s.reward, False, cat_uniform=u.data_ptr() if u is not None else None)
            self._keep_u = u
            # proto.py:190-191 encodes obs and next_obs again for the actor / critic: next_obs with the weights and the input of the reward
            # pass above — the same values, kept — and obs with the stepped encoder, the one pass left to make (4 of the update's 20
            # forward convolutions gone)
            eng.encode(0)
        stddev = self._stddev(step)
        eng.set_train_encoder(False)
        eng.update(stddev, None, None, self.noise_hook((B, A)) if self.noise_hook else None, self.noise_hook((B, A)) if self.noise_hook else None,
                   keep_augmented=not self.reward_free, keep_encoded=self.reward_free)
        eng.encoder_target(self.encoder_target_tau)
        metrics = dict()
        if self.use_tb or self.use_wandb:
            raw = eng.metrics_raw()
            for idx, name in _CRITIC_METRICS + [(L.M_ACTOR_LOGPROB, 'actor_logprob')]:
                metrics[name] = float(raw[idx])
            metrics['actor_ent'] = float(np.float32(0.5 + 0.5 * np.log(2 * np.pi) + np.log(stddev)) * self.action_dim)
            if self.reward_free:
                ri = self.intr.metrics_raw()
                metrics['repr_loss'] = float(ri[L.IM_LOSS])
                metrics['intr_reward'] = float(ri[L.IM_INTR_REWARD])
                metrics['extr_reward'] = float(ri[L.IM_EXTR_REWARD])
        return metrics

    def update(self, replay_iter, step):
        if self.obs_type == 'pixels':
            if step % self.update_every_steps != 0:
                return dict()
            return self._update_pixels(replay_iter, step)
        return super().update(replay_iter, step)


_ENC_KEYS = [f'convnet.{i}.{w}' for i in (0, 2, 4, 6) for w in ('weight', 'bias')]
_PIX_ACTOR_KEYS = ['trunk.0.weight', 'trunk.0.bias', 'trunk.1.weight', 'trunk.1.bias'] + [f'policy.{i}.{w}' for i in (0, 2, 4) for w in ('weight', 'bias')]
_PIX_CRITIC_KEYS = (['trunk.0.weight', 'trunk.0.bias', 'trunk.1.weight', 'trunk.1.bias'] +
                    [f'{q}.{i}.{w}' for q in ('Q1', 'Q2') for i in (0, 2, 4) for w in ('weight', 'bias')])


class _ParamList(NetView):
    """A NetView over explicit tensors (encoder_target: a Polyak copy that lives outside the engine's net table)."""

    def __init__(self, keys, tensors):
        self._engine, self._net, self._keys, self._on_change = None, None, list(keys), None
        self.training = True
        self._params = list(tensors)


class _PixelNetView(NetView):
    """NetView over a PixelEngine net; conv weights are exposed in torch's (co, ci, 3, 3) shape."""

    def __init__(self, engine, net, keys, shapes=None):
        self._engine, self._net, self._keys, self._on_change = engine, net, list(keys), None
        self.training = True
        self._params = []
        for i in range(len(keys)):
            t = engine.tensor(net, i, L.T_PARAM)
            self._params.append(t.view(*shapes[i]) if shapes else t)

    def grads(self):
        return [self._engine.tensor(self._net, i, L.T_GRAD) for i in range(len(self._keys))]


def _pixel_init(c_in, hw, A, F, H, meta_dim=0, sf_dim=0):
    """Initial tensors of Encoder, pixel Actor, pixel Critic (and the critic_target's discarded draws) in the reference's RNG order
    (ddpg.py:165-181): every module default-initialised at construction, then weight_init — orthogonal with the ReLU gain for
    Conv2d, gain 1 for Linear, zero biases (utils.py:59-69)."""
    def orth(m, gain=1.0):
        nn.init.orthogonal_(m.weight.data, gain)
        if m.bias is not None:
            m.bias.data.fill_(0.0)
    convs = [nn.Conv2d(c_in, 32, 3, stride=2)] + [nn.Conv2d(32, 32, 3, stride=1) for _ in range(3)]
    for m in convs:
        orth(m, nn.init.calculate_gain('relu'))
    e = (hw - 3) // 2 + 1 - 6
    R = 32 * e * e

    def net(head_in, out, n_heads):
        mods = [nn.Linear(R + meta_dim, F), nn.LayerNorm(F)]
        for _ in range(n_heads):
            mods += [nn.Linear(head_in, H), nn.Linear(H, H), nn.Linear(H, out)]
        for m in mods:
            if isinstance(m, nn.Linear):
                orth(m)
        return [t for m in mods for t in (m.weight.data, m.bias.data)]
    actor = net(F, A, 1)
    critic = net(F + A, 1, 2)
    net(F + A, 1, 2)                      # critic_target's draws, overwritten by load_state_dict
    if sf_dim:                            # APSAgent replaces both with CriticSF after DDPG's constructor ran (aps.py:94-104)
        critic = net(F + A, sf_dim, 2)
        net(F + A, sf_dim, 2)
    return {'encoder': [t for m in convs for t in (m.weight.data, m.bias.data)], 'actor': actor, 'critic': critic, 'repr_dim': R}


class _Identity:
    training = True

    def train(self, mode=True):
        self.training = mode
        return self

    def parameters(self):
        return []

    def __call__(self, x):
        return x
