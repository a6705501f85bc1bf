"""Generates tests/golden/* by running the REFERENCE hot path (read-only /root/reference) on CPU.

Run in the build container only:  python tools/gen_golden.py [--only replay|utils|tiny|full]
The outputs are data (inputs + the reference's outputs); no reference source is copied.
Inputs that are seed-regenerable live in tests/_synth.py and are NOT stored in the fixtures.

What each fixture pins:
  replay_*.npz     G1: replay_buffer.py:153-277 (index streams, eviction order, n-step values)
  utils_g2.npz     G2: utils.py single ops (TruncatedNormal, schedule, soft update, RMS, PBE) + Adam
  tiny_<agent>.npz G3: 5 update() steps at H=32,B=8 with explicit weights, noise, metrics, final params
  full_<agent>.json G4: 10 update() steps at BASELINE dims, scalar metrics + parameter checksums,
                        fp32 (1 thread) and an fp64 adjudication run
"""
import argparse
import json
import os
import random
import sys
import tempfile
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
sys.path.insert(0, str(HERE.parent / 'tests'))
from ref_harness import load_reference  # noqa: E402
import _synth  # noqa: E402

GOLD = HERE.parent / 'tests' / 'golden'


# ----------------------------------------------------------------------------- replay (G1)
def gen_replay(ref):
    rb = ref.rb
    scenarios = {
        # name: (lengths, obs_dim, act_dim, meta_dim, nstep, max_size, batch, nbatches, seed, obs_u8)
        'a_nstep1': ([7, 3, 12, 5, 9, 4, 3, 15, 6, 8, 10, 3, 11], 5, 2, 0, 1, 10**6, 16, 4, 3, False),
        'b_nstep3': ([7, 3, 12, 5, 9, 4, 3, 15, 6, 8, 10, 3, 11], 5, 2, 0, 3, 10**6, 16, 4, 4, False),
        'c_evict': ([7, 3, 12, 5, 9, 4, 3, 15, 6, 8, 10, 3, 11], 5, 2, 0, 3, 40, 16, 4, 5, False),
        'd_meta': ([6, 9, 4, 11, 5], 4, 3, 2, 2, 10**6, 8, 3, 6, False),
        'e_single': ([5], 3, 1, 0, 5, 10**6, 8, 2, 7, False),   # one episode, len == nstep
        'f_pixels': ([6, 4, 9], 48, 2, 0, 3, 10**6, 8, 2, 8, True),  # uint8 observations
    }
    for name, (lengths, O, A, M, nstep, max_size, B, NB, seed, u8) in scenarios.items():
        eps = _synth.synth_episodes(seed, lengths, O, A, M, u8)
        with tempfile.TemporaryDirectory() as td:
            specs = (ref.Array((O,), np.uint8 if u8 else np.float32, 'observation'),
                     ref.Array((A,), np.float32, 'action'),
                     ref.Array((1,), np.float32, 'reward'),
                     ref.Array((1,), np.float32, 'discount'))
            meta_specs = (ref.Array((M,), np.float32, 'skill'),) if M else tuple()
            storage = rb.ReplayBufferStorage(specs, meta_specs, Path(td) / 'buffer')
            for ep in eps:
                storage._store_episode(ep)          # replay_buffer.py:143-150
            assert len(storage) == sum(lengths)
            loader = rb.make_replay_loader(storage, max_size, B, 0, True, nstep, 0.99)
            picks, starts = [], []
            orig_se = rb.ReplayBuffer._sample_episode
            orig_ri = np.random.randint

            def rec_se(self):
                fn = random.choice(self._episode_fns)
                picks.append(int(fn.stem.split('_')[1]))
                return self._episodes[fn]

            def rec_ri(*a, **k):
                v = orig_ri(*a, **k)
                starts.append(int(v) + 1)
                return v

            rb.ReplayBuffer._sample_episode = rec_se
            np.random.randint = rec_ri
            try:
                random.seed(seed)
                np.random.seed(seed)
                it = iter(loader)
                batches = [next(it) for _ in range(NB)]
                resident = [int(fn.stem.split('_')[1]) for fn in loader.dataset._episode_fns]
            finally:
                rb.ReplayBuffer._sample_episode = orig_se
                np.random.randint = orig_ri
        out = dict(lengths=np.array(lengths), dims=np.array([O, A, M, nstep, max_size, B, NB, seed, int(u8)]),
                   picks=np.array(picks, np.int64), starts=np.array(starts, np.int64),
                   resident=np.array(resident, np.int64))
        for bi, b in enumerate(batches):
            for ti, t in enumerate(b):
                out[f'batch{bi}_{ti}'] = t.numpy()
        np.savez_compressed(GOLD / f'replay_{name}.npz', **out)
        print('replay', name, 'resident', resident, 'first picks', picks[:5], starts[:5])


# ----------------------------------------------------------------------------- offline replay (R5)
class _FakePhysics:
    """Stands where dm_control's Physics stands in relabel_episode (replay_buffer.py:31-42): holds the state it was given."""

    def __init__(self):
        self.state = None

    def reset_context(self):
        import contextlib
        return contextlib.nullcontext()

    def set_state(self, s):
        self.state = np.array(s, np.float64)


class _FakeTask:
    def get_reward(self, physics):           # any deterministic function of the physics state
        return float(np.tanh(physics.state).sum() * 0.25)


class FakeEnv:
    def __init__(self, ref):
        self.physics, self.task, self._ref = _FakePhysics(), _FakeTask(), ref

    def reward_spec(self):
        return self._ref.Array((1,), np.float32, 'reward')


def gen_offline(ref):
    """OfflineReplayBuffer (replay_buffer.py:45-100) through make_offline_replay_loader (:246-258), num_workers=0.
    As shipped the class cannot load (`_load` calls self._relable_reward, the method is named _relabel_reward), so:
      a/b: `_load(relable=False)` is called explicitly (the method's own parameter), then the loader is iterated;
      c:   the instance gets the missing alias (`_relable_reward = _relabel_reward`) and runs its default path with a fake env.
    No reference file is modified."""
    rb = ref.rb
    lengths = [7, 3, 12, 5, 9, 4, 3, 15, 6, 8, 10, 3, 11]
    scenarios = {'offline_a': (10**6, 3, False), 'offline_b_cap': (40, 4, False), 'offline_c_relabel': (10**6, 5, True)}
    for name, (max_size, seed, relabel) in scenarios.items():
        O, A, B, NB, P = 5, 2, 16, 4, 3
        eps = _synth.synth_episodes(seed, lengths, O, A)
        rs = np.random.RandomState(100 + seed)
        for ep in eps:
            ep['physics'] = rs.standard_normal((ep['observation'].shape[0], P))
        with tempfile.TemporaryDirectory() as td:
            d = Path(td) / 'buffer'
            d.mkdir()
            for i, ep in enumerate(eps):
                rb.save_episode(ep, d / f'episode_{i}_{ep["observation"].shape[0] - 1}.npz')
            env = FakeEnv(ref)
            loader = rb.make_offline_replay_loader(env, d, max_size, B, 0, 0.99)
            ds = loader.dataset
            if relabel:
                ds._relable_reward = ds._relabel_reward
            else:
                ds._load(relable=False)
                ds._loaded = True
            random.seed(seed)
            np.random.seed(seed)
            it = iter(loader)
            batches = [next(it) for _ in range(NB)]
            resident = [int(fn.stem.split('_')[1]) for fn in ds._episode_fns]
            out = dict(lengths=np.array(lengths), dims=np.array([O, A, P, max_size, B, NB, seed, int(relabel)]),
                       resident=np.array(resident, np.int64), size=np.array(ds._size))
            if relabel:
                for i in resident:
                    out[f'reward{i}'] = ds._episodes[d / f'episode_{i}_{lengths[i]}.npz']['reward']
            for bi, b in enumerate(batches):
                for ti, t in enumerate(b):
                    out[f'batch{bi}_{ti}'] = t.numpy()
            assert sorted(p.name for p in d.glob('*.npz')) == sorted(f'episode_{i}_{n}.npz' for i, n in enumerate(lengths))   # nothing deleted
        np.savez_compressed(GOLD / f'replay_{name}.npz', **out)
        print('offline', name, 'resident', resident, 'size', ds._size)


# ----------------------------------------------------------------------------- utils (G2)
def gen_utils(ref):
    U = ref.utils
    rs = np.random.RandomState(11)
    out = {}
    # TruncatedNormal.sample, utils.py:128-149 (noise recorded)
    mu = np.tanh(rs.standard_normal((6, 4)) * 1.5).astype(np.float32)
    noise = rs.standard_normal((6, 4)).astype(np.float32) * 2.0
    orig = U._standard_normal
    U._standard_normal = lambda shape, dtype, device: torch.from_numpy(noise.copy()).to(dtype)
    try:
        for tag, clip in (('clip', 0.3), ('noclip', None)):
            loc = torch.from_numpy(mu.copy()).requires_grad_(True)
            d = U.TruncatedNormal(loc, torch.ones_like(loc) * 0.2)
            x = d.sample(clip=clip)
            (x * torch.arange(24.).view(6, 4)).sum().backward()
            out[f'tn_{tag}_x'] = x.detach().numpy()
            out[f'tn_{tag}_grad'] = loc.grad.numpy()
        d = U.TruncatedNormal(torch.from_numpy(mu), torch.ones(6, 4) * 0.2)
        a = torch.from_numpy(rs.uniform(-1, 1, (6, 4)).astype(np.float32))
        out['tn_a'] = a.numpy()
        out['tn_logprob'] = d.log_prob(a).numpy()
        out['tn_entropy'] = d.entropy().numpy()
    finally:
        U._standard_normal = orig
    out['tn_mu'], out['tn_noise'] = mu, noise
    # schedule, utils.py:199-219
    sch = ['0.2', 'linear(1.0,0.1,100)', 'step_linear(1.0,0.5,50,0.1,100)']
    steps = [0, 10, 50, 75, 100, 1000]
    out['schedule'] = np.array([[U.schedule(s, t) for t in steps] for s in sch], np.float64)
    # soft update, utils.py:44-47
    net = torch.nn.Linear(7, 5)
    tgt = torch.nn.Linear(7, 5)
    out['soft_w'], out['soft_b'] = net.weight.detach().numpy().copy(), net.bias.detach().numpy().copy()
    out['soft_tw0'], out['soft_tb0'] = tgt.weight.detach().numpy().copy(), tgt.bias.detach().numpy().copy()
    U.soft_update_params(net, tgt, 0.01)
    out['soft_tw1'], out['soft_tb1'] = tgt.weight.detach().numpy().copy(), tgt.bias.detach().numpy().copy()
    # Adam (torch.optim.Adam defaults as td3_bc.py:96-97), 4 steps on explicit grads
    p = torch.nn.Parameter(torch.from_numpy(rs.standard_normal(33).astype(np.float32)))
    opt = torch.optim.Adam([p], lr=1e-4)
    out['adam_p0'] = p.detach().numpy().copy()
    grads = rs.standard_normal((4, 33)).astype(np.float32) * np.array([1, 1e-3, 10, 1e-6], np.float32)[:, None]
    out['adam_grads'] = grads
    ps = []
    for g in grads:
        opt.zero_grad(set_to_none=True)
        p.grad = torch.from_numpy(g.copy())
        opt.step()
        ps.append(p.detach().numpy().copy())
    out['adam_p'] = np.stack(ps)
    # RMS + PBE, utils.py:257-319
    rep = rs.standard_normal((16, 8)).astype(np.float32)
    out['pbe_rep'] = rep
    for tag, (avg, use_rms, clip, k) in dict(avg=(True, False, 0.0, 3), kth=(False, False, 0.0, 3),
                                            avg_rms=(True, True, 0.0005, 4), kth_rms_noclip=(False, True, -1.0, 2)).items():
        rms = U.RMS('cpu')
        pbe = U.PBE(rms, clip, k, avg, use_rms, 'cpu')
        r1 = pbe(torch.from_numpy(rep.copy()))
        r2 = pbe(torch.from_numpy(rep.copy() * 1.5))        # second call exercises the running stats
        out[f'pbe_{tag}_r1'], out[f'pbe_{tag}_r2'] = r1.numpy(), r2.numpy()
        out[f'pbe_{tag}_M'], out[f'pbe_{tag}_S'] = rms.M.numpy(), rms.S.numpy()
    rms = U.RMS('cpu')
    xs = rs.standard_normal((3, 10, 1)).astype(np.float32)
    out['rms_x'] = xs
    ms = [np.stack([t.numpy() for t in rms(torch.from_numpy(x))]) for x in xs]
    out['rms_MS'] = np.stack(ms)
    # Proto-style kNN reward (proto.py:114-119): distances to a queue, 3rd smallest
    z = rs.standard_normal((10, 6)).astype(np.float32)
    queue = rs.standard_normal((20, 6)).astype(np.float32)
    zt, qt = torch.from_numpy(z), torch.from_numpy(queue)
    z_to_q = torch.norm(zt[:, None, :] - qt[None, :, :], dim=2, p=2)
    all_dists, _ = torch.topk(z_to_q, 3, dim=1, largest=False)
    out['knn_z'], out['knn_queue'], out['knn_reward'] = z, queue, all_dists[:, -1:].numpy()
    # orthogonal init plumbing (utils.py:59-69) under a fixed torch seed
    torch.manual_seed(5)
    lin = torch.nn.Linear(5, 32)
    lin.apply(U.weight_init)
    lin2 = torch.nn.Linear(32, 3)
    lin2.apply(U.weight_init)
    out['init_w_5_32'], out['init_w_32_3'] = lin.weight.detach().numpy(), lin2.weight.detach().numpy()
    np.savez_compressed(GOLD / 'utils_g2.npz', **out)
    print('utils g2 keys', len(out))


# ----------------------------------------------------------------------------- pixels (G5: K14 encoder, K15 augmentation)
def gen_pixels(ref):
    """utils.RandomShiftsAug (utils.py:222-254) with explicit shifts, and ddpg.Encoder (ddpg.py:12-39) forward + backward."""
    U = ref.utils
    out = {}
    rs = np.random.RandomState(3)
    for tag, (n, c, h) in dict(small=(3, 2, 12), full=(2, 3, 84)).items():
        x = rs.randint(0, 256, (n, c, h, h)).astype(np.uint8)
        shifts = rs.randint(0, 9, (n, 1, 1, 2))
        shifts[0, 0, 0] = (0, 8)                                   # extremes of the range
        o_randint = torch.randint
        torch.randint = lambda lo, hi, size, device=None, dtype=None: torch.from_numpy(shifts.copy()).to(dtype)
        try:
            y = U.RandomShiftsAug(pad=4)(torch.from_numpy(x))
        finally:
            torch.randint = o_randint
        out[f'aug_{tag}_x'], out[f'aug_{tag}_shift'], out[f'aug_{tag}_y'] = x, shifts.reshape(n, 2).astype(np.int32), y.numpy()
    for tag, (n, c) in dict(c3=(2, 3), c9=(2, 9)).items():
        torch.manual_seed(17)
        enc = ref.ddpg.Encoder((c, 84, 84))
        x = rs.randint(0, 256, (n, c, 84, 84)).astype(np.uint8)
        xt = torch.from_numpy(x).float().requires_grad_(True)
        hfeat = enc(xt)
        dh = torch.from_numpy(np.random.RandomState(7).standard_normal(tuple(hfeat.shape)).astype(np.float32))
        (hfeat * dh).sum().backward()
        out[f'enc_{tag}_x'] = x
        for k, v in enc.state_dict().items():
            out[f'enc_{tag}_param/{k}'] = v.numpy().copy()
        for k, p in enc.named_parameters():
            out[f'enc_{tag}_grad/{k}'] = p.grad.numpy().copy()
        # the full feature map is 39200 floats per image: keep a strided sample + checksums (fixture size), all of dx for one image channel
        hf = hfeat.detach().numpy()
        out[f'enc_{tag}_h_sample'] = hf[:, ::97].copy()
        out[f'enc_{tag}_h_sums'] = np.array([hf.astype(np.float64).sum(), (hf.astype(np.float64) ** 2).sum()])
        out[f'enc_{tag}_dx0'] = xt.grad.numpy()[0, 0].copy()
    np.savez_compressed(GOLD / 'pixels_g5.npz', **out)
    print('pixels g5 keys', len(out))


def gen_pixel_ddpg(ref):
    """DDPGAgent with obs_type='pixels' (ddpg.py:126-328): 3 update() calls, B=4, (3,84,84) uint8 frames. Weights come from
    _synth.synth_params seeds (the 39200-wide trunks are too big to store); augmentation shifts and action noise are recorded."""
    U = ref.utils
    C, HW, A, F, H, B, N = 3, 84, 3, 16, 32, 4, 3
    torch.manual_seed(5)
    agent = ref.ddpg.DDPGAgent('ddpg', True, 'pixels', (C, HW, HW), (A,), 'cpu', 1e-4, F, H, 0.01, 2000, 2, 0.2, 3, B, 0.3, True, True, False)
    out = {'dims': np.array([C, HW, A, F, H, B, N])}
    for nm, net in (('encoder', agent.encoder), ('actor', agent.actor), ('critic', agent.critic)):      # initial weights under manual_seed(5)
        out[f'init_sums/{nm}'] = np.array([[float(v.double().sum()), float((v.double() ** 2).sum())] for v in net.state_dict().values()])
    for i, (nm, net) in enumerate((('encoder', agent.encoder), ('actor', agent.actor), ('critic', agent.critic))):
        shapes = [(k, tuple(v.shape)) for k, v in net.state_dict().items()]
        params = _synth.synth_params(shapes, 50 + i)
        net.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
        out[f'keys/{nm}'] = np.array([k for k, _ in shapes])
    agent.critic_target.load_state_dict(agent.critic.state_dict())
    rs = np.random.RandomState(8)
    shifts, noise = [], _synth.NoiseStream(21)
    o_sn, o_randint = U._standard_normal, torch.randint

    def p_randint(lo, hi, size, device=None, dtype=None):
        sh = rs.randint(lo, hi, tuple(size))
        shifts.append(sh.reshape(-1, 2).astype(np.int32))
        return torch.from_numpy(sh).to(dtype)
    U._standard_normal = lambda shape, dtype, device: torch.from_numpy(noise.draw(shape)).to(dtype)
    torch.randint = p_randint
    metrics = []
    try:
        for i in range(N):
            b = _synth.synth_batch(61, i, B, 4, A)
            obs = rs.randint(0, 256, (B, C, HW, HW)).astype(np.uint8)
            nobs = rs.randint(0, 256, (B, C, HW, HW)).astype(np.uint8)
            out[f'batch/{i}/obs'], out[f'batch/{i}/next_obs'] = obs, nobs
            out[f'batch/{i}/action'], out[f'batch/{i}/reward'], out[f'batch/{i}/discount'] = b[1], b[2], b[3]
            m = agent.update(iter([(obs, b[1], b[2], b[3], nobs)]), 2 * i)
            metrics.append({k: float(v) for k, v in m.items()})
    finally:
        U._standard_normal, torch.randint = o_sn, o_randint
    out['shifts'] = np.stack(shifts)                      # [2 * step + (0 obs | 1 next_obs)]
    keys = sorted(metrics[0].keys())
    out['metric_keys'] = np.array(keys)
    out['metrics'] = np.array([[m[k] for k in keys] for m in metrics], np.float64)
    for nm, net in (('encoder', agent.encoder), ('actor', agent.actor), ('critic', agent.critic), ('critic_target', agent.critic_target)):
        for k, v in net.state_dict().items():
            v = v.numpy()
            if v.size <= 20000:
                out[f'final/{nm}/{k}'] = v.copy()
            else:                                         # trunk.0.weight (F x 39200): strided sample + checksums
                out[f'final_sample/{nm}/{k}'] = v.reshape(-1)[::997].copy()
                out[f'final_sums/{nm}/{k}'] = np.array([v.astype(np.float64).sum(), (v.astype(np.float64) ** 2).sum()])
    np.savez_compressed(GOLD / 'pixel_ddpg.npz', **out)
    print('pixel ddpg', keys, out['metrics'][-1])


def gen_pixel_proto(ref):
    """ProtoAgent with obs_type='pixels' (proto.py:46-207; BASELINE config 4 in miniature): 3 update() calls, B=4."""
    import math
    import torch.distributions as pyd
    U = ref.utils
    C, HW, A, F, H, B, N = 3, 84, 3, 16, 32, 4, 3
    PD, PJ, Q, NP = 8, 16, 24, 6
    torch.manual_seed(5)
    agent = ref.proto.ProtoAgent(pred_dim=PD, proj_dim=PJ, queue_size=Q, num_protos=NP, tau=0.1, encoder_target_tau=0.05, topk=3, update_encoder=True,
                                 name='proto', reward_free=True, obs_type='pixels', obs_shape=(C, HW, HW), action_shape=(A,), device='cpu', lr=1e-4,
                                 feature_dim=F, hidden_dim=H, critic_target_tau=0.01, num_expl_steps=2000, update_every_steps=2,
                                 stddev_schedule=0.2, nstep=3, batch_size=B, stddev_clip=0.3, init_critic=True, use_tb=True, use_wandb=False)
    out = {'dims': np.array([C, HW, A, F, H, B, N, PD, PJ, Q, NP])}
    mods = (('encoder', agent.encoder), ('actor', agent.actor), ('critic', agent.critic), ('predictor', agent.predictor),
            ('projector', agent.projector), ('protos', agent.protos))
    for i, (nm, net) in enumerate(mods):
        shapes = [(k, tuple(v.shape)) for k, v in net.state_dict().items()]
        params = _synth.synth_params(shapes, 50 + i)
        net.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
    agent.critic_target.load_state_dict(agent.critic.state_dict())
    agent.encoder_target.load_state_dict(agent.encoder.state_dict())
    agent.predictor_target.load_state_dict(agent.predictor.state_dict())
    rs = np.random.RandomState(8)
    shifts, noise, us = [], _synth.NoiseStream(21), []
    o_sn, o_randint, o_cat = U._standard_normal, torch.randint, pyd.Categorical.sample

    def p_randint(lo, hi, size, device=None, dtype=None):
        sh = rs.randint(lo, hi, tuple(size))
        shifts.append(sh.reshape(-1, 2).astype(np.int32))
        return torch.from_numpy(sh).to(dtype)

    def p_cat(self, sample_shape=torch.Size()):
        u = rs.uniform(0, 1, self.probs.shape[0])
        us.append(u)
        cdf = torch.cumsum(self.probs.double(), dim=1).numpy()
        idx = [min(int(np.searchsorted(cdf[i], u[i] * cdf[i, -1], side='right')), cdf.shape[1] - 1) for i in range(len(u))]
        return torch.tensor(idx, dtype=torch.long)
    U._standard_normal = lambda shape, dtype, device: torch.from_numpy(noise.draw(shape)).to(dtype)
    torch.randint, pyd.Categorical.sample = p_randint, p_cat
    metrics = []
    try:
        for i in range(N):
            b = _synth.synth_batch(61, i, B, 4, A)
            obs = rs.randint(0, 256, (B, C, HW, HW)).astype(np.uint8)
            nobs = rs.randint(0, 256, (B, C, HW, HW)).astype(np.uint8)
            out[f'batch/{i}/obs'], out[f'batch/{i}/next_obs'] = obs, nobs
            out[f'batch/{i}/action'], out[f'batch/{i}/reward'], out[f'batch/{i}/discount'] = b[1], b[2], b[3]
            m = agent.update(iter([(obs, b[1], b[2], b[3], nobs)]), 2 * i)
            metrics.append({k: float(v) for k, v in m.items()})
    finally:
        U._standard_normal, torch.randint, pyd.Categorical.sample = o_sn, o_randint, o_cat
    out['shifts'], out['cat_uniform'] = np.stack(shifts), np.stack(us)
    keys = sorted(metrics[0].keys())
    out['metric_keys'] = np.array(keys)
    out['metrics'] = np.array([[m[k] for k in keys] for m in metrics], np.float64)
    for nm, net in mods + (('critic_target', agent.critic_target), ('encoder_target', agent.encoder_target), ('predictor_target', agent.predictor_target)):
        for k, v in net.state_dict().items():
            v = v.numpy()
            if v.size <= 20000:
                out[f'final/{nm}/{k}'] = v.copy()
            else:
                out[f'final_sample/{nm}/{k}'] = v.reshape(-1)[::997].copy()
    out['final/queue'], out['final/queue_ptr'] = agent.queue.numpy().copy(), np.array(agent.queue_ptr)
    np.savez_compressed(GOLD / 'pixel_proto.npz', **out)
    print('pixel proto', keys, out['metrics'][-1])


CONFIG4 = dict(C=3, HW=84, A=9, F=50, H=1024, B=1024, N=3, PD=128, PJ=512, Q=2048, NP=512)


def _run_config4(ref, threads, mode='fp32'):
    import time
    import torch.distributions as pyd
    U = ref.utils
    c = CONFIG4
    C, HW, A, F, H, B, N, PD, PJ, Q, NP = (c[k] for k in ('C', 'HW', 'A', 'F', 'H', 'B', 'N', 'PD', 'PJ', 'Q', 'NP'))
    torch.set_num_threads(threads)
    torch.manual_seed(5)
    agent = ref.proto.ProtoAgent(pred_dim=PD, proj_dim=PJ, queue_size=Q, num_protos=NP, tau=0.1, encoder_target_tau=0.05, topk=3, update_encoder=True,
                                 name='proto', reward_free=True, obs_type='pixels', obs_shape=(C, HW, HW), action_shape=(A,), device='cpu', lr=1e-4,
                                 feature_dim=F, hidden_dim=H, critic_target_tau=0.01, num_expl_steps=2000, update_every_steps=2,
                                 stddev_schedule=0.2, nstep=3, batch_size=B, stddev_clip=0.3, init_critic=True, use_tb=True, use_wandb=False)
    params = _synth.config4_params(C, A, F, H, PD, PJ, NP)
    mods = (('encoder', agent.encoder), ('actor', agent.actor), ('critic', agent.critic), ('predictor', agent.predictor),
            ('projector', agent.projector), ('protos', agent.protos))
    for nm, net in mods:
        sd = net.state_dict()
        assert list(sd.keys()) == list(params[nm].keys()), (nm, list(sd.keys()))
        net.load_state_dict({k: torch.from_numpy(v).reshape(sd[k].shape) for k, v in params[nm].items()})
    agent.critic_target.load_state_dict(agent.critic.state_dict())
    agent.encoder_target.load_state_dict(agent.encoder.state_dict())
    agent.predictor_target.load_state_dict(agent.predictor.state_dict())
    if mode == 'fp64':      # the reference's own modules in double; its augmentation casts to float itself (utils.py:228), its output is widened
        for net in (agent.encoder, agent.encoder_target, agent.actor, agent.critic, agent.critic_target, agent.predictor, agent.predictor_target,
                    agent.projector, agent.protos):
            net.double()
        agent.queue = agent.queue.double()
        aug32 = agent.aug
        agent.aug = lambda x: aug32(x).double()
    noise = _synth.NoiseStream(22)
    o_sn, o_randint, o_cat = U._standard_normal, torch.randint, pyd.Categorical.sample
    state = {}

    def p_randint(lo, hi, size, device=None, dtype=None):
        sh = state['shifts'].pop(0)
        assert tuple(size) == (B, 1, 1, 2) and lo == 0 and hi == 9, (lo, hi, size)
        return torch.from_numpy(sh.reshape(tuple(size)).astype(np.int64)).to(dtype)

    def p_cat(self, sample_shape=torch.Size()):
        u = state['u'].astype(np.float64)
        cdf = torch.cumsum(self.probs.double(), dim=1).numpy()
        idx = [min(int(np.searchsorted(cdf[i], u[i] * cdf[i, -1], side='right')), cdf.shape[1] - 1) for i in range(len(u))]
        return torch.tensor(idx, dtype=torch.long)
    U._standard_normal = lambda shape, dtype, device: torch.from_numpy(noise.draw(shape)).to(dtype)
    torch.randint, pyd.Categorical.sample = p_randint, p_cat
    metrics = []
    import contextlib
    ctx = torch.backends.mkldnn.flags(enabled=False) if mode == 'fp32_no_onednn' else contextlib.nullcontext()
    try:
        with ctx:
            for i in range(N):
                obs, nobs, act, rew, disc, so, sn, u = _synth.config4_inputs(i, B, C, HW, A, NP)
                state['shifts'], state['u'] = [so, sn], u
                t0 = time.time()
                m = agent.update(iter([(obs, act, rew, disc, nobs)]), 2 * i)
                print(f'config4 reference ({mode}, {threads} threads) step {i}: {time.time() - t0:.1f} s', {k: round(float(v), 6) for k, v in m.items()}, flush=True)
                metrics.append({k: float(v) for k, v in m.items()})
    finally:
        U._standard_normal, torch.randint, pyd.Categorical.sample = o_sn, o_randint, o_cat
    return agent, mods, params, metrics


def gen_config4(ref):
    """BASELINE.json configs[3] AT ITS OWN SIZES, run through the reference itself: ProtoAgent(obs_type='pixels') on (3, 84, 84) uint8 frames,
    A = 9, feature_dim 50, hidden 1024, pred_dim 128, proj_dim 512, 512 prototypes, queue 2048, nstep 3, batch 1024 (configs/agent/proto.yaml +
    pretrain.yaml), three update() calls on the CPU — four times:
      metrics            fp32, all threads, oneDNN convolutions          (the trajectory the tests read as "the reference")
      metrics_1thread    fp32, one thread                                 (same kernels, other blocking of the partial sums)
      metrics_no_onednn  fp32, all threads, torch's native convolutions   (another fp32 summation order of the same arithmetic)
      metrics_fp64       the reference's modules in double                (what all of them approximate)
    The spread of the three fp32 runs around the fp64 one is the reference's OWN fp32 reproducibility band at this size: Adam's first steps
    move each of the 2 M trunk weights by lr * sign(g), so a gradient rounding difference becomes a full +-lr step wherever it flips a sign,
    and everything evaluated through a once-stepped network carries it (actor_loss, a cancelling mean, most of all). The tests derive their bar
    for those metrics from this band and hold the rest to 1e-4. Stored besides: every 997th element of every final / initial tensor (fp32 run and
    fp64 run), a queue sample. Frames, rows, shifts, uniforms and noise are regenerated from seeds (_synth.config4_inputs)."""
    nthr = os.cpu_count() or 8
    agent, mods, params, metrics = _run_config4(ref, nthr)
    _, _, _, metrics1 = _run_config4(ref, 1)
    _, _, _, metrics_n = _run_config4(ref, nthr, 'fp32_no_onednn')
    agent64, mods64, _, metrics64 = _run_config4(ref, nthr, 'fp64')
    c = CONFIG4
    out = {'dims': np.array([c[k] for k in ('C', 'HW', 'A', 'F', 'H', 'B', 'N', 'PD', 'PJ', 'Q', 'NP')])}
    keys = sorted(metrics[0].keys())
    out['metric_keys'] = np.array(keys)
    tab = lambda ms: np.array([[m[k] for k in keys] for m in ms], np.float64)
    out['metrics'], out['metrics_1thread'], out['metrics_no_onednn'], out['metrics_fp64'] = tab(metrics), tab(metrics1), tab(metrics_n), tab(metrics64)
    samp = lambda w: w.copy() if w.size <= 4096 else w.reshape(-1)[::997].copy()
    for tag, ag, md in (('', agent, mods), ('_fp64', agent64, mods64)):
        for nm, net in md + (('critic_target', ag.critic_target), ('encoder_target', ag.encoder_target), ('predictor_target', ag.predictor_target)):
            for k, v in net.state_dict().items():
                v = v.numpy()
                out[f'final_sample{tag}/{nm}/{k}'] = samp(v)
                if not tag:
                    out[f'init_sample/{nm}/{k}'] = samp(params[nm.replace('_target', '')][k].reshape(v.shape))
    out['final/queue_sample'], out['final/queue_ptr'] = agent.queue.numpy().reshape(-1)[::97].copy(), np.array(agent.queue_ptr)
    np.savez_compressed(GOLD / 'config4_proto_b1024.npz', **out)
    print('config4 proto', keys)
    for nm in ('metrics', 'metrics_1thread', 'metrics_no_onednn'):
        rel = np.abs(out[nm] - out['metrics_fp64']) / (np.abs(out['metrics_fp64']) + 1e-12)
        print(f'{nm:18s} vs fp64, worst relative difference per step:', rel.max(axis=1), [keys[j] for j in rel.argmax(axis=1)])


# ----------------------------------------------------------------------------- agents (G3/G4)
PIXEL_INTR = ('icm', 'icm_apt', 'disagreement', 'diayn', 'aps', 'smm', 'rnd')


def pixel_intr_frames(step, B, C, HW):
    """uint8 frames of step `step` (regenerated by the tests, not stored)."""
    rs = np.random.RandomState(1000 + step)
    return rs.randint(0, 256, (B, C, HW, HW)).astype(np.uint8), rs.randint(0, 256, (B, C, HW, HW)).astype(np.uint8)


def gen_pixel_intr(ref, kind):
    """The reward-free agents with an intrinsic-reward module on pixel observations (icm.py:94-139, icm_apt.py:112-158,
    disagreement.py:88-136, diayn.py:125-176): 3 update() calls, B=4, (3,84,84) uint8 frames. Weights from _synth.synth_params seeds
    (encoder 50, actor 51, critic 52, module 53); frames from pixel_intr_frames; augmentation shifts and action noise recorded."""
    U = ref.utils
    C, HW, A, F, H, B, N, S = 3, 84, 3, 16, 32, 4, 3, 4
    torch.manual_seed(5)
    kw = dict(name=kind, reward_free=True, obs_type='pixels', obs_shape=(C, HW, HW), action_shape=(A,), device='cpu', lr=1e-4, feature_dim=F,
              hidden_dim=H, critic_target_tau=0.01, num_expl_steps=2000, update_every_steps=2, stddev_schedule=0.2, nstep=3, batch_size=B,
              stddev_clip=0.3, init_critic=True, use_tb=True, use_wandb=False)
    if kind == 'icm':
        agent, mod = ref.icm.ICMAgent(icm_scale=1.0, update_encoder=True, **kw), 'icm'
    elif kind == 'icm_apt':
        agent, mod = ref.icm_apt.ICMAPTAgent(icm_scale=1.0, knn_rms=True, knn_k=3, knn_avg=True, knn_clip=0.0, update_encoder=True, icm_rep_dim=16, **kw), 'icm'
    elif kind == 'disagreement':
        agent, mod = ref.disagreement.DisagreementAgent(update_encoder=True, **kw), 'disagreement'
    elif kind == 'diayn':
        agent, mod = ref.diayn.DIAYNAgent(update_skill_every_step=50, skill_dim=S, diayn_scale=1.0, update_encoder=True, skill_type='uniform', **kw), 'diayn'
    elif kind == 'aps':
        agent, mod = ref.aps.APSAgent(update_task_every_step=5, sf_dim=S, knn_rms=True, knn_k=3, knn_avg=True, knn_clip=0.0001, num_init_steps=4096,
                                      lstsq_batch_size=4096, update_encoder=True, **kw), 'aps'
    elif kind == 'rnd':
        agent, mod = ref.rnd.RNDAgent(rnd_rep_dim=16, update_encoder=True, rnd_scale=1.0, **kw), 'rnd'
    elif kind == 'smm':
        agent, mod = ref.smm.SMMAgent(z_dim=S, sp_lr=1e-3, vae_lr=1e-4, vae_beta=0.5, state_ent_coef=1.0, latent_ent_coef=1.0, latent_cond_ent_coef=1.0,
                                      update_encoder=True, **kw), 'smm'
    else:
        raise ValueError(kind)
    out = {'dims': np.array([C, HW, A, F, H, B, N, S])}
    mods = (('encoder', agent.encoder), ('actor', agent.actor), ('critic', agent.critic), (mod, getattr(agent, mod)))
    for i, (nm, net) in enumerate(mods):
        shapes = [(k, tuple(v.shape)) for k, v in net.state_dict().items()]
        if nm == 'rnd':         # predictor.0 IS agent.encoder (already loaded); the Linear layers get seed 53, the frozen encoder copy seed 54
            lin = [(k, sh) for k, sh in shapes if len(sh) <= 2 and not k.startswith('normalize_obs') and '.0.convnet.' not in k]
            conv = [(k, sh) for k, sh in shapes if k.startswith('target.0.convnet.')]
            params = dict(_synth.synth_params(lin, 53))
            params.update(_synth.synth_conv_params(conv, 54))
            net.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=False)
            out[f'keys/{nm}'] = np.array([k for k, _ in lin + conv])
            continue
        params = (_synth.synth_conv_params if nm == 'encoder' else _synth.synth_params)(shapes, 50 + i)
        net.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
        out[f'keys/{nm}'] = np.array([k for k, _ in shapes])
    agent.critic_target.load_state_dict(agent.critic.state_dict())
    rs = np.random.RandomState(8)
    shifts, noise = [], _synth.NoiseStream(21)
    o_sn, o_randint = U._standard_normal, torch.randint

    def p_randint(lo, hi, size, device=None, dtype=None):
        sh = rs.randint(lo, hi, tuple(size))
        shifts.append(sh.reshape(-1, 2).astype(np.int32))
        return torch.from_numpy(sh).to(dtype)
    U._standard_normal = lambda shape, dtype, device: torch.from_numpy(noise.draw(shape)).to(dtype)
    torch.randint = p_randint
    o_randn, eps_stream = torch.randn, _synth.NoiseStream(33)
    if kind == 'smm':                                     # the VAE's epsilon (smm.py:62)
        torch.randn = lambda shape, *a, **k: torch.from_numpy(eps_stream.draw(tuple(shape)))
    metrics = []
    try:
        for i in range(N):
            b = _synth.synth_batch(61, i, B, 4, A)
            obs, nobs = pixel_intr_frames(i, B, C, HW)
            batch = [obs, b[1], b[2], b[3], nobs]
            if kind in ('diayn', 'smm'):
                skill = np.zeros((B, S), np.float32)
                skill[np.arange(B), np.random.RandomState(70 + i).randint(0, S, B)] = 1.0
                batch.append(skill)
                out[f'batch/{i}/skill'] = skill
            if kind == 'aps':                             # unit-norm task vectors (aps.py:122-128)
                task = np.random.RandomState(70 + i).standard_normal((B, S)).astype(np.float32)
                task /= np.linalg.norm(task, axis=1, keepdims=True)
                batch.append(task)
                out[f'batch/{i}/skill'] = task
            m = agent.update(iter([tuple(batch)]), 2 * i)
            metrics.append({k: float(v) for k, v in m.items()})
    finally:
        U._standard_normal, torch.randint, torch.randn = o_sn, o_randint, o_randn
    out['shifts'] = np.stack(shifts)                      # [2 * step + (0 obs | 1 next_obs)]
    keys = sorted(metrics[0].keys())
    out['metric_keys'] = np.array(keys)
    out['metrics'] = np.array([[m[k] for k in keys] for m in metrics], np.float64)
    for nm, net in mods + (('critic_target', agent.critic_target),):
        for k, v in net.state_dict().items():
            v = v.numpy()
            if v.size <= 20000:
                out[f'final/{nm}/{k}'] = v.copy()
            else:                                         # 39200-wide tensors: strided sample
                out[f'final_sample/{nm}/{k}'] = v.reshape(-1)[::997].copy()
    if hasattr(agent, 'pbe'):
        out['final/rms'] = np.array([float(agent.pbe.rms.M), float(agent.pbe.rms.S), float(agent.pbe.rms.n)])
    if hasattr(agent, 'intrinsic_reward_rms'):
        r = agent.intrinsic_reward_rms
        out['final/rms'] = np.array([float(r.M), float(r.S), float(r.n)])
    np.savez_compressed(GOLD / f'pixel_{kind}.npz', **out)
    print('pixel', kind, keys, out['metrics'][-1])


def make_agent(ref, kind, O, A, H, B, device='cpu', use_tb=True, **kw):
    if kind == 'td3_bc':
        return ref.td3_bc.TD3BCAgent('td3_bc', (O,), (A,), device, 1e-4, H, 0.01, 0.2, 1, B, 0.3, use_tb, 2.5)
    if kind == 'td3':
        return ref.td3.TD3Agent('td3', (O,), (A,), device, 1e-4, H, 0.01, 0.2, 1, B, 0.3, use_tb)
    if kind == 'bc':
        return ref.bc.BCAgent('bc', (O,), (A,), device, 1e-4, H, B, 0.2, use_tb)
    if kind == 'cql':
        return ref.cql.CQLAgent('cql', (O,), (A,), device, 1e-4, H, 0.01, 1, B, use_tb, kw.get('alpha', 0.01), 3, 5.0,
                                kw.get('use_critic_lagrange', False))
    if kind == 'crr':
        return ref.crr.CRRAgent('crr', (O,), (A,), device, 1e-4, H, 0.01, 10, kw.get('weight_func', 'indicator'), 0.2, 1, B, 0.3, use_tb)
    ddpg_kw = dict(name=kind, reward_free=True, obs_type='states', obs_shape=(O,), action_shape=(A,), device=device, lr=1e-4,
                   feature_dim=50, hidden_dim=H, critic_target_tau=0.01, num_expl_steps=2000, update_every_steps=2,
                   stddev_schedule=0.2, nstep=3, batch_size=B, stddev_clip=0.3, init_critic=True, use_tb=use_tb, use_wandb=False)
    if kind == 'ddpg':
        return ref.ddpg.DDPGAgent(**ddpg_kw)
    if kind == 'rnd':          # configs/agent/rnd.yaml: rnd_rep_dim 512, rnd_scale 1.0 (tiny: 16)
        return ref.rnd.RNDAgent(rnd_rep_dim=kw.get('rep_dim', 16), update_encoder=True, rnd_scale=1.0, **ddpg_kw)
    if kind == 'icm':          # configs/agent/icm.yaml: icm_scale 1.0
        return ref.icm.ICMAgent(icm_scale=1.0, update_encoder=True, **ddpg_kw)
    if kind == 'proto':        # configs/agent/proto.yaml: pred_dim 128, proj_dim 512, queue 2048, 512 protos, tau 0.1, topk 3 (tiny: 8/16/24/6)
        return ref.proto.ProtoAgent(pred_dim=kw.get('pred_dim', 8), proj_dim=kw.get('proj_dim', 16), queue_size=kw.get('queue_size', 24),
                                    num_protos=kw.get('num_protos', 6), tau=0.1, encoder_target_tau=0.05, topk=3, update_encoder=True,
                                    **ddpg_kw)
    if kind == 'smm':          # configs/agent/smm.yaml: z_dim = skill_dim (4), sp_lr 1e-3, vae_lr 1e-2, vae_beta 0.5; coefficients pretrain.yaml
        return ref.smm.SMMAgent(z_dim=4, sp_lr=1e-3, vae_lr=1e-2, vae_beta=0.5, state_ent_coef=1.0, latent_ent_coef=1.0,
                                latent_cond_ent_coef=1.0, update_encoder=True, **ddpg_kw)
    if kind == 'aps':          # configs/agent/aps.yaml: sf_dim 10, knn_k 12, knn_avg true, knn_rms true, knn_clip 0.0001 (tiny: 4 / 3)
        return ref.aps.APSAgent(update_task_every_step=5, sf_dim=kw.get('sf_dim', 4), knn_rms=True, knn_k=kw.get('knn_k', 3), knn_avg=True,
                                knn_clip=0.0001, num_init_steps=4096, lstsq_batch_size=4096, update_encoder=True, **ddpg_kw)
    if kind == 'disagreement':
        return ref.disagreement.DisagreementAgent(update_encoder=True, **ddpg_kw)
    if kind == 'diayn':        # configs/agent/diayn.yaml: skill_dim 16, diayn_scale 1.0, update_skill_every_step 50 (tiny: 4 skills)
        return ref.diayn.DIAYNAgent(update_skill_every_step=50, skill_dim=kw.get('skill_dim', 4), diayn_scale=1.0,
                                    update_encoder=True, skill_type='uniform', **ddpg_kw)
    if kind == 'icm_apt':      # configs/agent/icm_apt.yaml: icm_rep_dim 512, knn_rms true, knn_k 12, knn_avg true, knn_clip 0.0
        return ref.icm_apt.ICMAPTAgent(icm_scale=1.0, knn_rms=kw.get('knn_rms', True), knn_k=kw.get('knn_k', 3),
                                       knn_avg=kw.get('knn_avg', True), knn_clip=kw.get('knn_clip', 0.0), update_encoder=True,
                                       icm_rep_dim=kw.get('rep_dim', 16), **ddpg_kw)
    raise ValueError(kind)


def nets_of(agent):
    nets = [('actor', agent.actor)]
    # (CQL's log_actor_alpha / log_critic_alpha scalars are stored separately by gen_tiny)
    if hasattr(agent, 'critic'):
        nets += [('critic', agent.critic), ('critic_target', agent.critic_target)]
    for nm in ('rnd', 'icm', 'disagreement', 'diayn', 'aps', 'smm', 'predictor', 'predictor_target', 'projector', 'protos'):            # intrinsic-reward modules of the DDPG-backbone agents
        if hasattr(agent, nm):
            nets.append((nm, getattr(agent, nm)))
    return nets


def run_agent(ref, agent, kind, nsteps, batch_fn, noise, dtype):
    """Drives agent.update() exactly as train_offline.py:114 / pretrain.py:280-283 do."""
    U = ref.utils
    orig = U._standard_normal
    U._standard_normal = lambda shape, dtype, device: torch.from_numpy(noise.draw(shape)).to(dtype)
    # CQL draws through three other doors (cql.py:159,170-176,238): torch.normal (Normal.sample), Tensor.uniform_ and
    # torch.distributions.normal._standard_normal (Normal.rsample). Route them all to the same deterministic stream;
    # uniform draws are U(-1,1) made from the stream's normals by the probability integral transform.
    import math
    import torch.distributions.normal as tdn
    o_normal, o_uniform, o_sn = torch.normal, torch.Tensor.uniform_, tdn._standard_normal

    def p_normal(mean, std, *a, **k):
        z = torch.from_numpy(noise.draw(tuple(mean.shape))).to(mean.dtype)
        return mean + std * z

    def p_uniform(self, lo=0.0, hi=1.0, **k):
        z = noise.draw(tuple(self.shape)).astype(np.float64)
        u = 0.5 * (1.0 + np.vectorize(math.erf)(z / math.sqrt(2.0)))
        self.copy_(torch.from_numpy(lo + (hi - lo) * u).to(self.dtype))
        return self
    import torch.distributions as pyd
    o_cat = pyd.Categorical.sample

    def p_cat(self, sample_shape=torch.Size()):
        # Categorical(prob).sample() (proto.py:103) by inverse CDF on uniforms made from the deterministic stream
        z = noise.draw((self.probs.shape[0],)).astype(np.float64)
        u = 0.5 * (1.0 + np.vectorize(math.erf)(z / math.sqrt(2.0)))
        cdf = torch.cumsum(self.probs.double(), dim=1).numpy()
        idx = [min(int(np.searchsorted(cdf[i], u[i] * cdf[i, -1], side='right')), cdf.shape[1] - 1) for i in range(len(u))]
        return torch.tensor(idx, dtype=torch.long)
    if kind == 'proto':
        pyd.Categorical.sample = p_cat
    o_randn = torch.randn
    if kind == 'smm':            # VAE.loss draws epsilon with torch.randn (smm.py:62)
        torch.randn = lambda *shape, **k: torch.from_numpy(noise.draw(tuple(shape[0]) if isinstance(shape[0], (list, tuple)) else shape))
    if kind == 'cql':
        torch.normal, torch.Tensor.uniform_ = p_normal, p_uniform
        tdn._standard_normal = lambda shape, dtype, device: torch.from_numpy(noise.draw(shape)).to(dtype)
    metrics = []
    try:
        for i in range(nsteps):
            step = 2 * i if kind in UNSUP else i           # ddpg.py:302 update_every_steps=2
            batch = tuple(x.astype(dtype) for x in batch_fn(i))
            m = agent.update(iter([batch]), step)
            metrics.append({k: float(v) for k, v in m.items()})
    finally:
        U._standard_normal = orig
        torch.normal, torch.Tensor.uniform_, tdn._standard_normal = o_normal, o_uniform, o_sn
        pyd.Categorical.sample = o_cat
        torch.randn = o_randn
    return metrics


def checksums(agent):
    cs = {}
    for nm, net in nets_of(agent):
        flat = torch.cat([p.detach().double().reshape(-1) for p in net.parameters()])
        cs[nm] = [float(flat.sum()), float((flat * flat).sum()), float(flat.abs().max())]
    return cs


TINY_CQL_LAGRANGE = 'cql-lagrange'
UNSUP = ('ddpg', 'rnd', 'icm', 'icm_apt', 'disagreement', 'diayn', 'proto', 'aps', 'smm')
TINY_KINDS = ('td3_bc', 'td3', 'bc', 'ddpg', 'crr', 'crr-exp', 'crr-identity', 'cql', 'rnd', 'icm', 'icm_apt', 'icm_apt-kth', 'disagreement', 'diayn', 'proto', 'cql-lagrange', 'aps', 'smm')


def gen_tiny(ref):
    O, A, H, B, N = 5, 3, 32, 8, 5
    for kind in TINY_KINDS:
        torch.manual_seed(21)
        base, _, wf = kind.partition('-')
        extra = {'weight_func': wf} if (wf and base == 'crr') else {}
        if kind == 'icm_apt-kth':
            extra = {'knn_avg': False, 'knn_clip': 0.0005}
        if kind == TINY_CQL_LAGRANGE:
            extra = {'use_critic_lagrange': True}
        agent = make_agent(ref, base, O, A, H, B, **extra)
        out = {}
        intr_log = []
        if hasattr(agent, 'compute_intr_reward'):        # record the intrinsic reward each step feeds to the critic
            inner = agent.compute_intr_reward

            def rec_intr(*a, **k):
                r = inner(*a, **k)
                if isinstance(r, tuple):         # APS: (entropy reward, successor-feature reward) — the critic sees their sum
                    intr_log.append(sum(x.detach() for x in r).numpy().copy())
                else:
                    intr_log.append(r.detach().numpy().copy())
                return r
            agent.compute_intr_reward = rec_intr
        for nm, net in nets_of(agent):
            for k, v in net.state_dict().items():
                out[f'init/{nm}/{k}'] = v.numpy().copy()

        class Rec:
            def __init__(self):
                self.inner = _synth.NoiseStream(77)
                self.log = []

            def draw(self, shape):
                x = self.inner.draw(shape)
                self.log.append(x.copy())      # sample() scales the returned tensor in place (utils.py:145)
                return x
        rec = Rec()
        batches = [_synth.synth_batch(31, i, B, O, A) for i in range(N)]
        if base == 'smm':            # 6th batch element: one-hot z (smm.py:148-152)
            rsk = np.random.RandomState(47)
            batches = [b + (np.eye(4, dtype=np.float32)[rsk.randint(0, 4, B)],) for b in batches]
        if base == 'diayn':          # 6th batch element: the one-hot skill the replay buffer stores as meta (diayn.py:123-125)
            rsk = np.random.RandomState(41)
            batches = [b + (np.eye(4, dtype=np.float32)[rsk.randint(0, 4, B)],) for b in batches]
        if base == 'aps':            # 6th batch element: the task vector w, unit norm (aps.py:139-146)
            rsk = np.random.RandomState(43)
            ts = [rsk.standard_normal((B, 4)).astype(np.float32) for _ in batches]
            batches = [b + ((t / np.linalg.norm(t, axis=1, keepdims=True)).astype(np.float32),) for b, t in zip(batches, ts)]
        metrics = run_agent(ref, agent, base, N, lambda i: batches[i], rec, np.float32)
        for i, b in enumerate(batches):
            for j, t in enumerate(b):
                out[f'batch/{i}/{j}'] = t
        for i, x in enumerate(rec.log):
            out[f'noise/{i}'] = x
        keys = sorted(metrics[0].keys())
        out['metric_keys'] = np.array(keys)
        out['metrics'] = np.array([[m[k] for k in keys] for m in metrics], np.float64)
        for nm, net in nets_of(agent):
            for k, v in net.state_dict().items():
                out[f'final/{nm}/{k}'] = v.numpy().copy()
        if intr_log:
            out['intr_reward'] = np.stack(intr_log)
        if base == 'proto':
            out['final/queue'] = agent.queue.numpy().copy()
            out['final/queue_ptr'] = np.array(agent.queue_ptr)
        if intr_log and isinstance(intr_log[0], tuple):       # APS returns (entropy reward, successor-feature reward)
            pass
        rms = getattr(agent, 'intrinsic_reward_rms', None) or getattr(getattr(agent, 'pbe', None), 'rms', None)
        if rms is not None:
            out['final/rms'] = np.array([float(rms.M), float(rms.S), float(rms.n)], np.float64)
        if base == 'cql':
            out['final/log_actor_alpha'] = agent.log_actor_alpha.detach().numpy().copy()
            out['final/log_critic_alpha'] = agent.log_critic_alpha.detach().numpy().copy()
        np.savez_compressed(GOLD / f'tiny_{kind}.npz', **out)
        print('tiny', kind, keys, out['metrics'][-1])


FULL = {  # kind: (O, A, H, B)  — BASELINE.json configs (walker / cheetah shapes)
    'td3_bc': (24, 6, 1024, 1024),
    'td3': (17, 6, 1024, 1024),
    'bc': (24, 6, 1024, 256),
    'ddpg': (24, 6, 1024, 1024),
    'crr': (24, 6, 1024, 1024),
    'cql': (78, 12, 1024, 1024),         # quadruped shapes, BASELINE.json configs[2]
    'td3_b4096': (17, 6, 1024, 4096),    # cheetah shapes at the global batch of BASELINE.json configs[4]
}
FULL_STEPS = {'td3_b4096': 5}
SAMPLE_STRIDE = 997                      # every 997th element of each net's flat parameters is stored (final values)


def param_sample(agent):
    """Final parameters at a fixed stride: lets a test compare parameter *deltas* element by element (the update direction and
    size, not just norms) without shipping 13 MB of weights."""
    out = {}
    for nm, net in nets_of(agent):
        flat = torch.cat([p.detach().double().reshape(-1) for p in net.parameters()])
        out[nm] = [float(x) for x in flat[::SAMPLE_STRIDE]]
    return out


def gen_full(ref, nsteps=10, only_kinds=None):
    torch.set_num_threads(1)
    for name, (O, A, H, B) in FULL.items():
        if only_kinds and name not in only_kinds:
            continue
        kind = name.partition('_b')[0] if name.endswith('4096') else name
        nsteps = FULL_STEPS.get(name, 10)
        res = {'dims': [O, A, H, B], 'nsteps': nsteps, 'param_seed': 5, 'batch_seed': 9, 'noise_seed': 13, 'sample_stride': SAMPLE_STRIDE}
        for tag, dtype, tdt in (('fp32', np.float32, torch.float32), ('fp64', np.float64, torch.float64)):
            agent = make_agent(ref, kind, O, A, H, B)
            for nm, net in nets_of(agent):
                if nm == 'critic_target':
                    continue
                shapes = [(k, tuple(v.shape)) for k, v in net.state_dict().items()]
                params = _synth.synth_params(shapes, 5 + (0 if nm == 'actor' else 1))
                net.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
            if hasattr(agent, 'critic'):
                agent.critic_target.load_state_dict(agent.critic.state_dict())
            for nm, net in nets_of(agent):
                net.to(tdt)
            if kind == 'cql' and tdt == torch.float64:
                agent.log_actor_alpha = agent.log_actor_alpha.detach().double().requires_grad_(True)
                agent.actor_alpha_opt = torch.optim.Adam([agent.log_actor_alpha], lr=1e-4)
            metrics = run_agent(ref, agent, kind, nsteps, lambda i: _synth.synth_batch(9, i, B, O, A),
                                _synth.NoiseStream(13), dtype)
            res[tag] = {'metrics': metrics, 'checksums': checksums(agent), 'param_sample': param_sample(agent)}
            print('full', name, tag, metrics[-1])
        with open(GOLD / f'full_{name}.json', 'w') as f:
            json.dump(res, f, indent=1)


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--only', default=None)
    ap.add_argument('--kinds', default=None, help='comma list: restrict tiny/full generation to these agent kinds')
    args = ap.parse_args()
    GOLD.mkdir(parents=True, exist_ok=True)
    ref = load_reference()
    todo = [args.only] if args.only else ['replay', 'offline', 'utils', 'tiny', 'full', 'pixels', 'pixel_ddpg', 'pixel_proto', 'config4'] + [f'pixel_{k}' for k in PIXEL_INTR]
    kinds = args.kinds.split(',') if args.kinds else None
    if kinds:
        TINY_KINDS = tuple(k for k in TINY_KINDS if k.partition('-')[0] in kinds)
    for t in todo:
        if t == 'full':
            gen_full(ref, only_kinds=kinds)
        else:
            fns = {'replay': gen_replay, 'offline': gen_offline, 'utils': gen_utils, 'tiny': gen_tiny, 'pixels': gen_pixels, 'pixel_ddpg': gen_pixel_ddpg,
                   'pixel_proto': gen_pixel_proto, 'config4': gen_config4}
            if t.startswith('pixel_') and t[6:] in PIXEL_INTR:
                gen_pixel_intr(ref, t[6:])
            else:
                fns[t](ref)
