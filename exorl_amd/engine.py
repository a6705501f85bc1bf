"""Thin object wrappers over the C ABI handles (exorl_agent_t, exorl_replay_t).

torch is used here only as plumbing: it owns the device workspace (so parameters can be exposed as
torch tensors for state_dict / pickling / torch.distributed all-reduce) and provides the stream.
All arithmetic happens in libexorl_hip.so.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib as L

KIND = {'td3_bc': L.AGENT_TD3_BC, 'td3': L.AGENT_TD3, 'bc': L.AGENT_BC, 'ddpg': L.AGENT_DDPG, 'crr': L.AGENT_CRR, 'cql': L.AGENT_CQL,
        'aps': L.AGENT_APS}
PRECISION = {'fp32': L.PREC_F32, 'f32': L.PREC_F32, 'bf16': L.PREC_BF16, 'bf16x3': L.PREC_BF16X3, 'bf16x6': L.PREC_BF16X6}
METRIC_KEYS = {L.M_BATCH_REWARD: 'batch_reward', L.M_CRITIC_TARGET_Q: 'critic_target_q', L.M_CRITIC_Q1: 'critic_q1',
               L.M_CRITIC_Q2: 'critic_q2', L.M_CRITIC_LOSS: 'critic_loss', L.M_ACTOR_LOSS: 'actor_loss',
               L.M_ACTOR_LOGPROB: 'actor_logprob'}


def _require_gpu(device):
    if not torch.cuda.is_available():
        raise L.ExorlError('exorl_amd needs a HIP device (torch.cuda.is_available() is False); there is no CPU path')
    dev = torch.device(device)
    if dev.type != 'cuda':
        raise L.ExorlError(f"exorl_amd agents run on the GPU only; got device={device!r}")
    return dev


class AgentEngine:
    def __init__(self, kind, obs_dim, act_dim, hidden_dim, batch, lr=1e-4, tau=0.01, alpha=2.5, stddev_clip=0.3,
                 precision='fp32', world_size=1, seed=0, device='cuda', num_value_samples=10, weight_func='indicator',
                 n_samples=3, use_critic_lagrange=False, target_cql_penalty=5.0, sf_dim=0):
        self.lib = L.load()
        self.device = _require_gpu(device)
        self.kind = kind
        self.cfg = L.AgentCfg(KIND[kind], obs_dim, act_dim, hidden_dim, batch, PRECISION[precision], world_size, sf_dim,
                              lr, tau, alpha, stddev_clip if stddev_clip is not None else 0.0, seed, num_value_samples,
                              L.CRR_WEIGHT[weight_func], n_samples, int(bool(use_critic_lagrange)), target_cql_penalty, 0)
        self.obs_dim, self.act_dim, self.hidden_dim, self.batch = obs_dim, act_dim, hidden_dim, batch
        nbytes = self.lib.exorl_agent_workspace_bytes(C.byref(self.cfg))
        if nbytes == 0:
            raise L.ExorlError(self.lib.exorl_last_error().decode())
        with torch.cuda.device(self.device):
            self.workspace = torch.zeros(nbytes + 256, dtype=torch.uint8, device=self.device)
            base = self.workspace.data_ptr()
            self._ws_off = (-base) % 256
            handle = C.c_void_p()
            L.check(self.lib.exorl_agent_create(C.byref(self.cfg), base + self._ws_off, nbytes, C.byref(handle)))
        self.h = handle
        self._f32 = self.workspace[self._ws_off:self._ws_off + nbytes].view(torch.float32)
        self.has_critic = kind != 'bc'
        self.comm = None

    def __del__(self):
        h, self.h = getattr(self, 'h', None), None
        if h:
            self.lib.exorl_agent_destroy(h)

    # ---- views of library-laid-out memory as torch tensors ------------------------------------------
    def _view(self, ptr, numel):
        off = (ptr - self._f32.data_ptr()) // 4
        return self._f32[off:off + numel]

    def num_tensors(self, net):
        n = C.c_int32()
        L.check(self.lib.exorl_agent_num_tensors(self.h, net, C.byref(n)))
        return n.value

    def tensor(self, net, index, what=L.T_PARAM):
        p, r, c = C.c_void_p(), C.c_int64(), C.c_int64()
        L.check(self.lib.exorl_agent_tensor(self.h, net, index, what, C.byref(p), C.byref(r), C.byref(c)))
        v = self._view(p.value, r.value * c.value)
        return v.view(r.value, c.value) if c.value > 1 else v

    def tensor_shaped(self, net, index, shape, what=L.T_PARAM):
        p, r, c = C.c_void_p(), C.c_int64(), C.c_int64()
        L.check(self.lib.exorl_agent_tensor(self.h, net, index, what, C.byref(p), C.byref(r), C.byref(c)))
        assert int(np.prod(shape)) == r.value * c.value, (shape, r.value, c.value)
        return self._view(p.value, r.value * c.value).view(*shape)

    def flat(self, net, what=L.T_PARAM):
        p, n = C.c_void_p(), C.c_int64()
        L.check(self.lib.exorl_agent_flat(self.h, net, what, C.byref(p), C.byref(n)))
        return self._view(p.value, n.value)

    def stats(self):
        p, n = C.c_void_p(), C.c_int64()
        L.check(self.lib.exorl_agent_stats_buffer(self.h, C.byref(p), C.byref(n)))
        return self._view(p.value, n.value)

    def batch_slots(self):
        out = L.BatchOut()
        L.check(self.lib.exorl_agent_batch_slots(self.h, C.byref(out)))
        return out

    # ---- operations ----------------------------------------------------------------------------------
    def params_changed(self, sync_target=False):
        L.check(self.lib.exorl_agent_params_changed(self.h, int(sync_target), L.current_stream()))

    def set_batch(self, obs, action, reward, discount, next_obs):
        ts = [self._dev(x) for x in (obs, action, reward, discount, next_obs)]
        B = self.batch
        for t, n in zip(ts, (B * self.obs_dim, B * self.act_dim, B, B, B * self.obs_dim)):
            if t.numel() != n:
                raise L.ExorlError(f'batch tensor has {t.numel()} elements, expected {n}')
        L.check(self.lib.exorl_agent_set_batch(self.h, *[t.data_ptr() for t in ts], L.current_stream()))
        self._keep = ts

    def _dev(self, x):
        t = torch.as_tensor(x)
        if t.dtype != torch.float32:
            t = t.float()
        return t.to(self.device, non_blocking=True).contiguous()

    def update(self, stddev, noise_critic=None, noise_actor=None):
        nc = self._dev(noise_critic) if noise_critic is not None else None
        na = self._dev(noise_actor) if noise_actor is not None else None
        L.check(self.lib.exorl_agent_update(self.h, stddev, L.ptr(nc), L.ptr(na), L.current_stream()))
        self._keep_noise = (nc, na)

    def update_phase(self, phase, stddev, noise_critic=None, noise_actor=None):
        nc = self._dev(noise_critic) if noise_critic is not None else None
        na = self._dev(noise_actor) if noise_actor is not None else None
        L.check(self.lib.exorl_agent_update_phase(self.h, phase, stddev, L.ptr(nc), L.ptr(na), L.current_stream()))
        self._keep_noise = (nc, na)

    def act(self, obs, stddev, eval_mode, noise=None):
        o = self._dev(obs).view(-1, self.obs_dim)
        n = o.shape[0]
        out = torch.empty(n, self.act_dim, dtype=torch.float32, device=self.device)
        nz = self._dev(noise) if noise is not None else None
        L.check(self.lib.exorl_agent_act(self.h, o.data_ptr(), n, stddev, int(eval_mode), L.ptr(nz), out.data_ptr(),
                                         L.current_stream()))
        return out

    def act_host(self, obs, stddev, eval_mode, noise=None):
        """act() for ONE observation in one kernel launch (exorl_agent_act_host): the row and the optional noise row travel as kernel
        arguments, the action lands in a pinned host slot. Returns a numpy (act_dim,) array, or None when the fused kernel does not apply
        (CQL's tanh-Gaussian policy, hidden_dim % 4 != 0) and the caller should take act()."""
        if self.kind == 'cql' or self.hidden_dim % 4 != 0 or self.obs_dim > 256:
            return None
        slot = getattr(self, '_act_slot', None)
        if slot is None:
            slot = self._act_slot = torch.zeros(64, dtype=torch.float32).pin_memory()
        o = np.ascontiguousarray(np.asarray(obs, np.float32).reshape(-1))
        assert o.size == self.obs_dim, (o.size, self.obs_dim)
        nz = None if noise is None else np.ascontiguousarray(np.asarray(noise, np.float32).reshape(-1))
        stream = L.current_stream()
        L.check(self.lib.exorl_agent_act_host(self.h, o.ctypes.data, 1, float(stddev), int(eval_mode), nz.ctypes.data if nz is not None else None,
                                              slot.data_ptr(), stream))
        torch.cuda.current_stream(self.device).synchronize()
        return slot[:self.act_dim].numpy().copy()

    def set_comm(self, comm):
        """Attach an exorl_amd.comm.Comm of cfg.world_size ranks: update() then runs the data-parallel step in one call."""
        L.check(self.lib.exorl_agent_set_comm(self.h, comm.h if comm is not None else None))
        self.comm = comm

    def enable_graph(self, replay_engine, nstep, gamma, stddev):
        L.check(self.lib.exorl_agent_enable_graph(self.h, replay_engine.h, nstep, gamma, stddev, L.current_stream()))
        self.graph_captures = getattr(self, 'graph_captures', 0) + 1

    def disable_graph(self):
        L.check(self.lib.exorl_agent_disable_graph(self.h))

    def step_graph(self, stddev):
        L.check(self.lib.exorl_agent_step_graph(self.h, stddev, L.current_stream()))

    def noise_counter(self):
        """Philox draw counter of the update noise AFTER the steps enqueued so far (synchronises): step k of a DDPG-family agent
        drew its critic-target noise at counter 2k+2 and its actor noise at 2k+3 (the counter is advanced at the step's start)."""
        c = C.c_uint64()
        L.check(self.lib.exorl_agent_noise_counter(self.h, C.byref(c), L.current_stream()))
        return int(c.value)

    def philox_normal(self, seed, counter, shape):
        """The standard-normal block the update kernels draw for (seed, counter): element e = row * A + column (test hook)."""
        out = torch.empty(tuple(shape), dtype=torch.float32, device=self.device)
        L.check(self.lib.exorl_debug_philox_normal(seed, counter, out.numel(), out.data_ptr(), L.current_stream()))
        return out

    def set_parallel_branches(self, enable):
        L.check(self.lib.exorl_agent_set_parallel_branches(self.h, int(bool(enable))))

    def set_metrics(self, enable):
        L.check(self.lib.exorl_agent_set_metrics(self.h, int(bool(enable))))

    def cql_alpha_state(self):
        host = np.zeros(6, np.float32)
        L.check(self.lib.exorl_agent_cql_alpha(self.h, host.ctypes.data, 0))
        return host            # log_actor_alpha, Adam m, Adam v [, log_critic_alpha, m, v with use_critic_lagrange]

    def set_cql_alpha_state(self, log_alpha, m=0.0, v=0.0, log_critic_alpha=0.0, cm=0.0, cv=0.0):
        host = np.array([log_alpha, m, v, log_critic_alpha, cm, cv], np.float32)
        L.check(self.lib.exorl_agent_cql_alpha(self.h, host.ctypes.data, 1))

    def metrics_raw(self):
        host = np.zeros(L.N_METRICS, np.float32)
        L.check(self.lib.exorl_agent_metrics(self.h, host.ctypes.data, L.current_stream()))
        return host

    def opt_steps(self):
        a, c = C.c_int64(), C.c_int64()
        L.check(self.lib.exorl_agent_opt_steps(self.h, C.byref(a), C.byref(c)))
        return a.value, c.value

    def set_opt_steps(self, actor_steps, critic_steps):
        L.check(self.lib.exorl_agent_set_opt_steps(self.h, actor_steps, critic_steps))


class IntrEngine:
    """Intrinsic-reward module (exorl_intr_t): RND / ICM / ICM-APT. Parameters live in a torch-owned workspace so they
    can be exposed as tensors (state_dict, snapshots)."""
    KINDS = {'rnd': L.INTR_RND, 'icm': L.INTR_ICM, 'icm_apt': L.INTR_ICM_APT, 'disagreement': L.INTR_DISAGREEMENT, 'diayn': L.INTR_DIAYN,
             'proto': L.INTR_PROTO, 'aps': L.INTR_APS, 'smm': L.INTR_SMM}

    def __init__(self, kind, obs_dim, act_dim, hidden_dim, batch, rep_dim=0, lr=1e-4, scale=1.0, knn_k=12, knn_avg=True,
                 knn_rms=True, knn_clip=0.0, clip_val=5.0, n_models=0, num_protos=0, queue_size=0, tau=0.1, target_tau=0.05, sp_lr=1e-3, vae_lr=1e-2,
                 vae_beta=0.5, state_ent_coef=1.0, latent_ent_coef=1.0, latent_cond_ent_coef=1.0, goal=(150.0, 75.0), precision='fp32',
                 device='cuda', encoded=False):
        self.lib = L.load()
        self.device = _require_gpu(device)
        self.kind, self.batch, self.obs_dim, self.act_dim = kind, batch, obs_dim, act_dim
        self.cfg = L.IntrCfg(self.KINDS[kind], obs_dim, act_dim, hidden_dim, rep_dim, batch, PRECISION[precision], knn_k, int(bool(knn_avg)),
                             int(bool(knn_rms)), n_models, 1 if encoded else 0, lr, scale, knn_clip, clip_val, num_protos, queue_size, tau, target_tau,
                             sp_lr, vae_lr, vae_beta, state_ent_coef, latent_ent_coef, latent_cond_ent_coef, goal[0], goal[1])
        nbytes = self.lib.exorl_intr_workspace_bytes(C.byref(self.cfg))
        if nbytes == 0:
            raise L.ExorlError(self.lib.exorl_last_error().decode())
        with torch.cuda.device(self.device):
            self.workspace = torch.zeros(nbytes + 256, dtype=torch.uint8, device=self.device)
            base = self.workspace.data_ptr()
            off = (-base) % 256
            handle = C.c_void_p()
            L.check(self.lib.exorl_intr_create(C.byref(self.cfg), base + off, nbytes, C.byref(handle)))
        self.h = handle
        self._f32 = self.workspace[off:off + nbytes].view(torch.float32)
        rms, bn, nbn = C.c_void_p(), C.c_void_p(), C.c_int64()
        L.check(self.lib.exorl_intr_state(self.h, C.byref(rms), C.byref(bn), C.byref(nbn)))
        self._rms = self._view(rms.value, 4)                       # {float M, float S, double n}
        self.bn = self._view(bn.value, nbn.value) if bn.value else None
        self.queue = None
        if kind == 'proto':
            q, r, c, ptr = C.c_void_p(), C.c_int64(), C.c_int64(), C.c_int64()
            L.check(self.lib.exorl_intr_queue(self.h, C.byref(q), C.byref(r), C.byref(c), C.byref(ptr), 0))
            self.queue = self._view(q.value, r.value * c.value).view(r.value, c.value)

    def queue_ptr(self, set_to=None):
        q, r, c = C.c_void_p(), C.c_int64(), C.c_int64()
        ptr = C.c_int64(0 if set_to is None else int(set_to))
        L.check(self.lib.exorl_intr_queue(self.h, C.byref(q), C.byref(r), C.byref(c), C.byref(ptr), int(set_to is not None)))
        return ptr.value

    def __del__(self):
        h, self.h = getattr(self, 'h', None), None
        if h:
            self.lib.exorl_intr_destroy(h)

    def _view(self, ptr, numel):
        off = (ptr - self._f32.data_ptr()) // 4
        return self._f32[off:off + numel]

    def num_tensors(self, net=None):
        n = C.c_int32()
        L.check(self.lib.exorl_intr_num_tensors(self.h, C.byref(n)))
        return n.value

    def tensor(self, net, index, what=L.T_PARAM):
        p, r, c = C.c_void_p(), C.c_int64(), C.c_int64()
        L.check(self.lib.exorl_intr_tensor(self.h, index, what, C.byref(p), C.byref(r), C.byref(c)))
        v = self._view(p.value, r.value * c.value)
        return v.view(r.value, c.value) if c.value > 1 else v

    def flat(self, what=L.T_PARAM):
        p, n = C.c_void_p(), C.c_int64()
        L.check(self.lib.exorl_intr_flat(self.h, what, C.byref(p), C.byref(n)))
        return self._view(p.value, n.value)

    def rms_state(self):
        """(M, S, n) of the module's utils.RMS."""
        raw = self._rms.cpu().numpy()
        return float(raw[0]), float(raw[1]), float(raw[2:4].view(np.float64)[0])

    def set_rms_state(self, M, S, n):
        raw = np.zeros(4, np.float32)
        raw[0], raw[1] = M, S
        raw[2:4] = np.array([n], np.float64).view(np.float32)
        self._rms.copy_(torch.from_numpy(raw))

    def update(self, obs, action, next_obs, extr_reward, reward_out, train=True, skill=None, obs_ld=None, action_ld=None,
               next_obs_ld=None, skill_ld=0, cat_uniform=None, next_obs_target=None, dobs_out=None):
        """Device pointers (ints) + row strides in floats; see exorl_intr_batch. train: True/1, False/0, or 2 (optimiser step only)."""
        b = L.IntrBatch(obs, obs_ld or self.obs_dim, action, action_ld or self.act_dim, next_obs, next_obs_ld or self.obs_dim,
                        skill, skill_ld, extr_reward, reward_out, next_obs_target, self.obs_dim, dobs_out, cat_uniform)
        L.check(self.lib.exorl_intr_update(self.h, C.byref(b), 2 if train == 2 else int(bool(train)), L.current_stream()))

    def metrics_raw(self):
        host = np.zeros(L.N_INTR_METRICS, np.float32)
        L.check(self.lib.exorl_intr_metrics(self.h, host.ctypes.data, L.current_stream()))
        return host

    def opt_steps(self):
        n = C.c_int64()
        L.check(self.lib.exorl_intr_opt_steps(self.h, C.byref(n), 0))
        return n.value

    def set_opt_steps(self, n):
        v = C.c_int64(n)
        L.check(self.lib.exorl_intr_opt_steps(self.h, C.byref(v), 1))

    def counter(self, set_to=None):
        c = C.c_uint64(0 if set_to is None else int(set_to))
        L.check(self.lib.exorl_intr_counter(self.h, C.byref(c), 0 if set_to is None else 1))
        return int(c.value)


class PixelEngine:
    """DDPG on pixel observations (exorl_pixel_agent_t): augmentation, conv encoder, pixel actor/critic and their update."""
    NETS = {'encoder': 0, 'actor': 1, 'critic': 2, 'critic_target': 3}

    def __init__(self, obs_shape, act_dim, feature_dim, hidden_dim, batch, lr=1e-4, tau=0.01, stddev_clip=0.3, precision='fp32', seed=0,
                 device='cuda', meta_dim=0, sf_dim=0):
        self.lib = L.load()
        self.device = _require_gpu(device)
        c, h, w = obs_shape
        if h != w:
            raise L.ExorlError(f'pixel observations must be square, got {obs_shape}')
        self.obs_shape, self.act_dim, self.batch, self.meta_dim = tuple(obs_shape), act_dim, batch, meta_dim
        self.cfg = L.PixelCfg(c, h, act_dim, feature_dim, hidden_dim, batch, PRECISION[precision], meta_dim, lr, tau,
                              stddev_clip if stddev_clip is not None else 0.0, sf_dim, seed)
        nbytes = self.lib.exorl_pixel_agent_workspace_bytes(C.byref(self.cfg))
        if nbytes == 0:
            raise L.ExorlError(self.lib.exorl_last_error().decode())
        with torch.cuda.device(self.device):
            self.workspace = torch.zeros(nbytes + 256, dtype=torch.uint8, device=self.device)
            base = self.workspace.data_ptr()
            off = (-base) % 256
            handle = C.c_void_p()
            L.check(self.lib.exorl_pixel_agent_create(C.byref(self.cfg), base + off, nbytes, C.byref(handle)))
        self.h = handle
        self._f32 = self.workspace[off:off + nbytes].view(torch.float32)

    def __del__(self):
        h, self.h = getattr(self, 'h', None), None
        if h:
            self.lib.exorl_pixel_agent_destroy(h)

    def _view(self, ptr, numel):
        off = (ptr - self._f32.data_ptr()) // 4
        return self._f32[off:off + numel]

    def num_tensors(self, net):
        n = C.c_int32()
        L.check(self.lib.exorl_pixel_agent_num_tensors(self.h, net, C.byref(n)))
        return n.value

    def tensor(self, net, index, what=L.T_PARAM):
        p, r, c = C.c_void_p(), C.c_int64(), C.c_int64()
        L.check(self.lib.exorl_pixel_agent_tensor(self.h, net, index, what, C.byref(p), C.byref(r), C.byref(c)))
        v = self._view(p.value, r.value * c.value)
        return v.view(r.value, c.value) if c.value > 1 else v

    def sync_target(self):
        L.check(self.lib.exorl_pixel_agent_sync_target(self.h, L.current_stream()))

    def batch_slots(self):
        out = L.BatchOut()
        L.check(self.lib.exorl_pixel_agent_batch_slots(self.h, C.byref(out)))
        return out

    def _u8(self, x):
        t = torch.as_tensor(x)
        if t.dtype != torch.uint8:
            raise L.ExorlError(f'pixel observations must be uint8 (got {t.dtype}), as the replay buffer stores them')
        return t.to(self.device, non_blocking=True).contiguous()

    def _f(self, x):
        return torch.as_tensor(x).to(self.device, torch.float32, non_blocking=True).contiguous()

    def set_batch(self, obs, action, reward, discount, next_obs):
        ts = [self._u8(obs), self._f(action), self._f(reward), self._f(discount), self._u8(next_obs)]
        L.check(self.lib.exorl_pixel_agent_set_batch(self.h, *[t.data_ptr() for t in ts], L.current_stream()))
        self._keep = ts

    def _i32(self, x):
        return None if x is None else torch.as_tensor(np.ascontiguousarray(x, np.int32)).to(self.device)

    def update(self, stddev, shifts_obs=None, shifts_next=None, noise_critic=None, noise_actor=None, keep_augmented=False, keep_encoded=False):
        f32 = lambda x: None if x is None else self._f(x)
        ts = [self._i32(shifts_obs), self._i32(shifts_next), f32(noise_critic), f32(noise_actor)]
        ptrs = [L.ptr(t) for t in ts]
        if keep_encoded:                    # reuse the encodings of the last encode(0) / encode(1) (sentinel pointer, see the header)
            ptrs[0] = C.c_void_p(-2)
        elif keep_augmented:                # reuse the images exorl_pixel_agent_augment made
            ptrs[0] = C.c_void_p(-1)
        L.check(self.lib.exorl_pixel_agent_update(self.h, stddev, *ptrs, L.current_stream()))
        self._keep_u = ts

    def augment(self, shifts_obs=None, shifts_next=None):
        ts = [self._i32(shifts_obs), self._i32(shifts_next)]
        L.check(self.lib.exorl_pixel_agent_augment(self.h, L.ptr(ts[0]), L.ptr(ts[1]), L.current_stream()))
        self._keep_s = ts

    def encode(self, which, target=False):
        """Device pointer of the (batch, repr_dim) features of the augmented obs (which=0) / next_obs (1)."""
        p = C.c_void_p()
        L.check(self.lib.exorl_pixel_agent_encode(self.h, which, int(bool(target)), C.byref(p), L.current_stream()))
        return p.value

    def encoder_step(self, which, dfeat_ptr, opt):
        L.check(self.lib.exorl_pixel_agent_encoder_step(self.h, which, dfeat_ptr, opt, L.current_stream()))

    def encoder_target(self, tau=0.0, init=False):
        L.check(self.lib.exorl_pixel_agent_encoder_target(self.h, tau, int(bool(init)), L.current_stream()))

    def encoder_target_tensors(self, shapes):
        p = C.c_void_p()
        L.check(self.lib.exorl_pixel_agent_encoder_target_ptr(self.h, C.byref(p)))
        out, off = [], 0
        for shp in shapes:
            n = int(np.prod(shp))
            out.append(self._view(p.value + 4 * off, n).view(*shp))
            off += (n + 3) // 4 * 4
        return out

    def set_train_encoder(self, enable):
        L.check(self.lib.exorl_pixel_agent_set_train_encoder(self.h, int(bool(enable))))

    # -- pickling support: everything that defines the training state, as CPU data
    def export_state(self):
        torch.cuda.synchronize()
        steps, ctr = np.zeros(3, np.int64), np.zeros(4, np.uint64)
        L.check(self.lib.exorl_pixel_agent_state(self.h, steps.ctypes.data, ctr.ctypes.data))
        st = {'steps': steps, 'counters': ctr, 'tensors': {}, 'bn2d': self.bn2d().cpu()}
        for net in range(4):
            for what in ((L.T_PARAM,) if net == 3 else (L.T_PARAM, L.T_ADAM_M, L.T_ADAM_V)):
                st['tensors'][(net, what)] = [self.tensor(net, i, what).cpu() for i in range(self.num_tensors(net))]
        m, v, n, p = C.c_void_p(), C.c_void_p(), C.c_int64(), C.c_void_p()
        L.check(self.lib.exorl_pixel_agent_encoder_opt2(self.h, C.byref(m), C.byref(v), C.byref(n)))
        L.check(self.lib.exorl_pixel_agent_encoder_target_ptr(self.h, C.byref(p)))
        st['enc_extra'] = [self._view(q.value, n.value).cpu() for q in (m, v, p)]
        return st

    def import_state(self, st):
        for (net, what), ts in st['tensors'].items():
            for i, t in enumerate(ts):
                self.tensor(net, i, what).copy_(t.reshape(self.tensor(net, i, what).shape))
        m, v, n, p = C.c_void_p(), C.c_void_p(), C.c_int64(), C.c_void_p()
        L.check(self.lib.exorl_pixel_agent_encoder_opt2(self.h, C.byref(m), C.byref(v), C.byref(n)))
        L.check(self.lib.exorl_pixel_agent_encoder_target_ptr(self.h, C.byref(p)))
        for q, t in zip((m, v, p), st['enc_extra']):
            self._view(q.value, n.value).copy_(t)
        steps, ctr = np.ascontiguousarray(st['steps'], np.int64), np.ascontiguousarray(st['counters'], np.uint64)
        if steps.size == 2:                 # ABI-6 pickles: encoder_opt stepped with the other optimisers
            steps = np.array([steps[0], steps[1], steps[0]], np.int64)
        if ctr.size == 3:
            ctr = np.concatenate([ctr, np.zeros(1, np.uint64)])
        if 'bn2d' in st:
            self.bn2d().copy_(st['bn2d'])
        L.check(self.lib.exorl_pixel_agent_set_state(self.h, steps.ctypes.data, ctr.ctypes.data))
        torch.cuda.synchronize()

    def metrics_raw(self):
        host = np.zeros(L.N_METRICS, np.float32)
        L.check(self.lib.exorl_pixel_agent_metrics(self.h, host.ctypes.data, L.current_stream()))
        return host

    def act(self, obs, stddev, eval_mode, noise=None, meta=None):
        o = self._u8(obs)
        out = torch.empty(self.act_dim, dtype=torch.float32, device=self.device)
        nz = self._f(noise) if noise is not None else None
        mt = self._f(meta) if meta is not None else None
        L.check(self.lib.exorl_pixel_agent_act(self.h, o.data_ptr(), L.ptr(mt), stddev, int(eval_mode), L.ptr(nz), out.data_ptr(), L.current_stream()))
        return out

    def bn2d(self):
        """RND's BatchNorm2d buffers: running_mean[c], running_var[c], num_batches_tracked (float) as one device view."""
        p, n = C.c_void_p(), C.c_int64()
        L.check(self.lib.exorl_pixel_agent_bn_state(self.h, C.byref(p), C.byref(n)))
        return self._view(p.value, n.value)

    def rnd_features(self, shifts=None, clip_val=5.0):
        """(predictor-side, target-side) encodings of clamp(BatchNorm2d(aug(obs))) as device pointers (rnd.py:47-53)."""
        sh = self._i32(shifts)
        fp, ft = C.c_void_p(), C.c_void_p()
        L.check(self.lib.exorl_pixel_agent_rnd_features(self.h, L.ptr(sh), clip_val, C.byref(fp), C.byref(ft), L.current_stream()))
        self._keep_r = sh
        return fp.value, ft.value

    def meta_rows(self):
        """(batch, meta_dim) device view of the skill / task rows the trunks read (filled by the sampler or by the caller)."""
        out = self.batch_slots()
        return self._view(out.meta, self.batch * self.meta_dim).view(self.batch, self.meta_dim)

    def feature_view(self, ptr):
        """(batch, repr_dim) tensor over the device pointer encode() returned."""
        n = self.lib.exorl_encoder_out_dim(self.obs_shape[1])
        return self._view(ptr, self.batch * n).view(self.batch, n)


class ReplayEngine:
    """HBM-resident episodic arena (exorl_replay_t)."""

    def __init__(self, obs_shape, obs_dtype, act_dim, meta_dim, capacity_rows, max_episodes, device='cuda'):
        self.lib = L.load()
        self.device = _require_gpu(device)
        self.obs_shape = tuple(obs_shape)
        self.obs_dtype = np.dtype(obs_dtype)
        self.obs_bytes = int(np.prod(self.obs_shape)) * self.obs_dtype.itemsize
        self.act_dim, self.meta_dim = act_dim, meta_dim
        cfg = L.ReplayCfg(self.obs_bytes, act_dim, meta_dim, max_episodes, capacity_rows)
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            L.check(self.lib.exorl_replay_create(C.byref(cfg), C.byref(h)))
        self.h = h

    def __del__(self):
        h, self.h = getattr(self, 'h', None), None
        if h:
            self.lib.exorl_replay_destroy(h)

    def append_episode(self, ep, meta_keys=()):
        obs = np.ascontiguousarray(ep['observation'])
        rows = obs.shape[0]
        assert obs.dtype == self.obs_dtype and obs.reshape(rows, -1).shape[1] * obs.itemsize == self.obs_bytes
        act = np.ascontiguousarray(ep['action'], np.float32).reshape(rows, -1)
        rew = np.ascontiguousarray(ep['reward'], np.float32).reshape(rows)
        disc = np.ascontiguousarray(ep['discount'], np.float32).reshape(rows)
        meta = None
        if self.meta_dim:
            meta = np.ascontiguousarray(np.concatenate([np.asarray(ep[k], np.float32).reshape(rows, -1) for k in meta_keys], 1))
            assert meta.shape[1] == self.meta_dim
        slot = C.c_int32()
        L.check(self.lib.exorl_replay_append_episode(self.h, obs.ctypes.data, act.ctypes.data, rew.ctypes.data,
                                                     disc.ctypes.data, L.ptr(meta), rows, C.byref(slot)))
        return slot.value

    def evict(self, slot):
        L.check(self.lib.exorl_replay_evict(self.h, slot))

    def num_rows(self):
        """(live rows, used rows) of the arena; a row is one time-step, an episode holds len+1."""
        live, used = C.c_int64(), C.c_int64()
        L.check(self.lib.exorl_replay_num_rows(self.h, C.byref(live), C.byref(used)))
        return live.value, used.value

    def set_order(self, slots):
        arr = np.ascontiguousarray(slots, np.int32)
        L.check(self.lib.exorl_replay_set_order(self.h, arr.ctypes.data, len(arr)))

    def seed_mt_from_globals(self):
        """Adopts the CURRENT state of Python's `random` and NumPy's legacy global generator — the two
        streams replay_buffer.py:169,222 draw from — so the index stream continues exactly where the
        reference's would (valid for num_workers=0; workers reseed unreproducibly, replay_buffer.py:241-244)."""
        import random
        st = random.getstate()[1]
        py_key, py_pos = np.array(st[:-1], np.uint32), int(st[-1])
        ns = np.random.get_state()
        np_key, np_pos = np.ascontiguousarray(ns[1], np.uint32), int(ns[2])
        L.check(self.lib.exorl_replay_seed_mt(self.h, py_key.ctypes.data, py_pos, np_key.ctypes.data, np_pos))

    def seed_mt_ints(self, py_seed, np_seed):
        L.check(self.lib.exorl_replay_seed_mt_ints(self.h, py_seed, np_seed))

    def seed_philox(self, seed):
        L.check(self.lib.exorl_replay_seed_philox(self.h, seed))

    def sample_into(self, out, batch, nstep, gamma, sampler, pairs=None, want_pairs=False):
        pin = np.ascontiguousarray(pairs, np.int32) if pairs is not None else None
        pout = np.zeros((batch, 2), np.int32) if want_pairs else None
        L.check(self.lib.exorl_replay_sample(self.h, batch, nstep, gamma, sampler, L.ptr(pin), C.byref(out), L.ptr(pout),
                                             L.current_stream()))
        return pout

    def last_pairs(self, batch):
        out = np.zeros((batch, 2), np.int32)
        L.check(self.lib.exorl_replay_last_pairs(self.h, batch, out.ctypes.data, L.current_stream()))
        return out

    def sample(self, batch, nstep, gamma, sampler=L.SAMPLER_MT19937, pairs=None, want_pairs=False):
        """Allocates fresh output tensors (B,*obs_shape) (B,A) (B,1) (B,1) (B,*obs_shape) [(B,meta)]."""
        tdt = torch.uint8 if self.obs_dtype == np.uint8 else torch.float32
        dev = self.device
        obs = torch.empty((batch,) + self.obs_shape, dtype=tdt, device=dev)
        nobs = torch.empty_like(obs)
        act = torch.empty(batch, self.act_dim, dtype=torch.float32, device=dev)
        rew = torch.empty(batch, 1, dtype=torch.float32, device=dev)
        disc = torch.empty(batch, 1, dtype=torch.float32, device=dev)
        meta = torch.empty(batch, self.meta_dim, dtype=torch.float32, device=dev) if self.meta_dim else None
        out = L.BatchOut(obs.data_ptr(), self.obs_bytes, act.data_ptr(), self.act_dim, rew.data_ptr(), disc.data_ptr(),
                         nobs.data_ptr(), self.obs_bytes, L.ptr(meta), self.meta_dim)
        p = self.sample_into(out, batch, nstep, gamma, sampler, pairs, want_pairs)
        res = (obs, act, rew, disc, nobs) + ((meta,) if meta is not None else ())
        return (res, p) if want_pairs else res
