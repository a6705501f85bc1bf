"""GPU parity of the HBM replay path through the C ABI: bit-exact against the reference's recorded
batches (tests/golden/replay_*.npz), against the oracle on seeded inputs, and by properties at full size."""
import random

import numpy as np
import pytest
import torch

import _synth
from oracle.replay import OracleReplay, episode_name, gather_nstep_batch, philox_draw

pytestmark = pytest.mark.gpu
SCEN = ['a_nstep1', 'b_nstep3', 'c_evict', 'd_meta', 'e_single', 'f_pixels']


class Spec:
    def __init__(self, shape, dtype, name):
        self.shape, self.dtype, self.name = tuple(shape), np.dtype(dtype), name


def write_dir(tmp_path, eps, meta_dim):
    from exorl_amd.replay_buffer import ReplayBufferStorage
    ms = (Spec((meta_dim,), np.float32, 'skill'),) if meta_dim else ()
    st = ReplayBufferStorage((), ms, tmp_path / 'buffer')
    for ep in eps:
        st._store_episode(ep)
    return st


@pytest.mark.parametrize('name', SCEN)
def test_golden_batches_bit_exact(gold, tmp_path, name):
    """make_replay_loader(...) with the global RNGs seeded like the reference run -> identical batches."""
    from exorl_amd.replay_buffer import make_replay_loader
    z = np.load(gold / f'replay_{name}.npz')
    O, A, M, nstep, max_size, B, NB, seed, u8 = [int(x) for x in z['dims']]
    lengths = [int(x) for x in z['lengths']]
    st = write_dir(tmp_path, _synth.synth_episodes(seed, lengths, O, A, M, bool(u8)), M)
    loader = make_replay_loader(st, max_size, B, 0, True, nstep, 0.99)
    random.seed(seed)
    np.random.seed(seed)
    it = iter(loader)
    for bi in range(NB):
        batch = next(it)
        for ti, t in enumerate(batch):
            ref = z[f'batch{bi}_{ti}']
            got = t.cpu().numpy()
            assert got.dtype == ref.dtype and got.shape == ref.shape, (name, bi, ti, got.shape, ref.shape)
            assert np.array_equal(got.view(np.uint8), ref.view(np.uint8)), (name, bi, ti)
    resident = [int(fn.stem.split('_')[1]) for fn in it.shards[0].fns]
    assert resident == list(z['resident'])


@pytest.mark.parametrize('name', ['offline_a', 'offline_b_cap', 'offline_c_relabel'])
def test_offline_loader_golden_batches_bit_exact(gold, tmp_path, name):
    """The 6-argument offline call shape (train_offline.py:90-93) -> OfflineReplayBuffer semantics (replay_buffer.py:45-100):
    ascending one-shot load with the `size > max_size` stop, relabelled rewards, nstep=1 samples — the reference's own batches,
    bit for bit; and the dataset directory is left exactly as it was."""
    from exorl_amd.replay_buffer import make_replay_loader, save_episode
    from test_oracle_replay import FakeEnv, offline_episodes
    z = np.load(gold / f'replay_{name}.npz')
    O, A, P, max_size, B, NB, seed, relabel = [int(x) for x in z['dims']]
    eps, lengths = offline_episodes(z)
    d = tmp_path / 'buffer'
    d.mkdir()
    for i, ep in enumerate(eps):
        save_episode(ep, d / f'episode_{i}_{lengths[i]}.npz')
    before = sorted((p.name, p.stat().st_size) for p in d.glob('*.npz'))
    loader = make_replay_loader(FakeEnv(), d, max_size, B, 0, 0.99, relabel=bool(relabel), sampler='mt19937')
    random.seed(seed)
    np.random.seed(seed)
    it = iter(loader)
    for bi in range(NB):
        batch = next(it)
        assert len(batch) == 5
        for ti, t in enumerate(batch):
            ref = z[f'batch{bi}_{ti}']
            got = t.cpu().numpy()
            assert got.dtype == ref.dtype and got.shape == ref.shape, (name, bi, ti, got.shape, ref.shape)
            assert np.array_equal(got.view(np.uint8), ref.view(np.uint8)), (name, bi, ti)
    sh = it.shards[0]
    assert [int(fn.stem.split('_')[1]) for fn in sh.fns] == list(z['resident']) and sh.size == int(z['size'])
    assert sorted((p.name, p.stat().st_size) for p in d.glob('*.npz')) == before           # nothing evicted, unlinked or rewritten
    live = sh.engine.num_rows()[0]
    assert live == sh.size + len(sh.fns)                                                   # arena sized from the data, not max_size


def test_offline_loader_needs_env_for_relabel(tmp_path):
    from exorl_amd.replay_buffer import make_offline_replay_loader
    with pytest.raises(ValueError, match='relabel=True needs the env'):
        make_offline_replay_loader(None, tmp_path, 100, 4, 0, 0.99, relabel=True)


def test_index_stream_equals_reference_recording(gold, tmp_path):
    from exorl_amd.engine import ReplayEngine
    from exorl_amd import _lib as L
    z = np.load(gold / 'replay_b_nstep3.npz')
    O, A, M, nstep, max_size, B, NB, seed, u8 = [int(x) for x in z['dims']]
    lengths = [int(x) for x in z['lengths']]
    eps = _synth.synth_episodes(seed, lengths, O, A)
    eng = ReplayEngine((O,), np.float32, A, 0, 4096, 64)
    slots = {i: eng.append_episode(ep) for i, ep in enumerate(eps)}
    order = sorted(range(len(eps)), key=lambda i: episode_name(i, lengths[i]))     # lexicographic, replay_buffer.py:184
    eng.set_order([slots[i] for i in order])
    eng.seed_mt_ints(seed, seed)
    picks, starts = [], []
    for _ in range(NB):
        _, pairs = eng.sample(B, nstep, 0.99, L.SAMPLER_MT19937, want_pairs=True)
        picks += [order[p] for p in pairs[:, 0]]
        starts += list(pairs[:, 1])
    assert picks == list(z['picks']) and starts == list(z['starts'])


def test_error_behaviour(tmp_path):
    from exorl_amd.engine import ReplayEngine
    from exorl_amd import _lib as L
    eng = ReplayEngine((4,), np.float32, 2, 0, 64, 32)
    with pytest.raises(L.ExorlError, match='no resident episodes'):
        eng.sample(4, 1, 0.99, L.SAMPLER_PHILOX)
    ep = _synth.synth_episodes(0, [3], 4, 2)[0]
    s = eng.append_episode(ep)
    eng.set_order([s])
    eng.seed_mt_ints(1, 1)
    with pytest.raises(L.ExorlError, match='shorter than nstep'):
        eng.sample(4, 5, 0.99, L.SAMPLER_MT19937)              # ValueError in the reference (randint(0, <=0))
    with pytest.raises(L.ExorlError, match='out of range'):
        eng.sample(2, 1, 0.99, L.SAMPLER_GIVEN, pairs=np.array([[0, 1], [0, 4]], np.int32))
    with pytest.raises(L.ExorlError, match='arena full'):
        for _ in range(20):
            eng.append_episode(ep)


def test_compaction_keeps_contents():
    from exorl_amd.engine import ReplayEngine
    from exorl_amd import _lib as L
    lengths = [9, 5, 7, 6, 8, 4]
    eps = _synth.synth_episodes(2, lengths, 6, 2)
    eng = ReplayEngine((6,), np.float32, 2, 0, 40, 8)          # 45 rows needed in total -> forces eviction + compaction
    slots = [eng.append_episode(eps[i]) for i in range(4)]     # 10+6+8+7 = 31 rows
    eng.evict(slots[1]); eng.evict(slots[2])
    slots[4:] = [eng.append_episode(eps[4])]                   # 9 rows -> 40 used
    slots.append(eng.append_episode(eps[5]))                   # 5 rows: does not fit -> compaction
    live = [0, 3, 4, 5]
    eng.set_order([slots[i] if i < 4 else slots[i] for i in live])
    pairs = np.array([[p, 1 + (j % (lengths[live[p]] - 1))] for j in range(32) for p in range(4)], np.int32)
    out = eng.sample(len(pairs), 2, 0.99, L.SAMPLER_GIVEN, pairs=pairs)
    for b, (p, idx) in enumerate(pairs):
        ep = eps[live[p]]
        assert np.array_equal(out[0][b].cpu().numpy(), ep['observation'][idx - 1])
        assert np.array_equal(out[4][b].cpu().numpy(), ep['observation'][idx + 1])
        assert np.array_equal(out[1][b].cpu().numpy(), ep['action'][idx])


@pytest.mark.parametrize('nstep', [1, 3])
def test_full_size_properties(nstep):
    """BASELINE config 2 scale: 1000 episodes x 1000 steps (1 M transitions, walker dims) in HBM, B=1024.
    Philox index stream == oracle restatement; values == flat-arena oracle gather, bit for bit."""
    from exorl_amd.engine import ReplayEngine
    from exorl_amd import _lib as L
    E, T, O, A, B = 1000, 1000, 24, 6, 1024
    rs = np.random.RandomState(1)
    eng = ReplayEngine((O,), np.float32, A, 0, E * (T + 1) + 16, E)
    obs = rs.standard_normal((E * (T + 1), O)).astype(np.float32)
    act = rs.uniform(-1, 1, (E * (T + 1), A)).astype(np.float32)
    rew = rs.uniform(0, 1, (E * (T + 1), 1)).astype(np.float32)
    disc = np.ones((E * (T + 1), 1), np.float32)
    disc[rs.uniform(size=len(disc)) < 0.01] = 0.0
    slots = []
    for e in range(E):
        sl = slice(e * (T + 1), (e + 1) * (T + 1))
        slots.append(eng.append_episode(dict(observation=obs[sl], action=act[sl], reward=rew[sl], discount=disc[sl])))
    eng.set_order(slots)
    eng.seed_philox(12345)
    lens = np.full(E, T)
    for counter in range(3):
        out = eng.sample(B, nstep, 0.99, L.SAMPLER_PHILOX)
        pairs = eng.last_pairs(B)
        want = np.array([philox_draw(12345, counter, b, E, lens, nstep) for b in range(B)])
        assert np.array_equal(pairs, want)
        ref = gather_nstep_batch(obs, act, rew, disc, pairs[:, 0].astype(np.int64) * (T + 1), pairs[:, 1].astype(np.int64),
                                 nstep, 0.99)
        for t, r in zip(out, ref):
            assert np.array_equal(t.cpu().numpy().view(np.uint8), np.ascontiguousarray(r).view(np.uint8))
        assert pairs[:, 1].min() >= 1 and pairs[:, 1].max() <= T - nstep + 1
    # distribution sanity: episodes and starts cover their ranges roughly uniformly
    big = np.concatenate([(eng.sample(B, nstep, 0.99, L.SAMPLER_PHILOX), eng.last_pairs(B))[1] for _ in range(20)])
    assert len(np.unique(big[:, 0])) > 0.99 * E
    assert abs(big[:, 1].mean() - (T - nstep + 2) / 2) < 0.05 * T


def test_in_process_hand_off_equals_the_file_path(tmp_path):
    """episodes finished in this process reach the HBM sampler from memory (storage._fresh) instead of being re-read and
    inflated from the file that was just written. Same batches, bit for bit, as a loader that only ever sees the files — including
    episodes added between batches (the reference's fetch cycle, replay_buffer.py:192-212)."""
    from exorl_amd.replay_buffer import ReplayBufferStorage, make_replay_loader
    O, A, B = 6, 2, 16
    eps = _synth.synth_episodes(21, [9, 14, 7, 11, 8, 12], O, A)
    live = ReplayBufferStorage((), (), tmp_path / 'buffer')
    for ep in eps[:3]:
        live._store_episode(ep)
    assert len(live._fresh) == 3
    cold = ReplayBufferStorage((), (), tmp_path / 'buffer')          # a second process' view: nothing in memory
    assert len(cold._fresh) == 0
    its = []
    for st in (live, cold):
        random.seed(4)
        np.random.seed(4)
        its.append(iter(make_replay_loader(st, 10**6, B, 0, True, 2, 0.99, fetch_every=B)))
    def same():
        a, b = next(its[0]), next(its[1])
        for x, y in zip(a, b):
            assert torch.equal(x, y)
    same()
    assert len(live._fresh) == 0                                      # taken over by the sampler
    for ep in eps[3:]:
        live._store_episode(ep)                                        # new episodes while sampling (cold sees them as files)
    for _ in range(4):
        same()
    assert [fn.name for fn in its[0].shards[0].fns] == [fn.name for fn in its[1].shards[0].fns] and len(its[0].shards[0].fns) == 6


def test_pixel_arena_past_4_gib_is_addressed_with_64_bit_offsets():
    """BASELINE config 4 keeps ~21 GB of uint8 frames in one arena: row * obs_bytes leaves 32 bits after row 202,900. A 4.7 GB arena
    of (3, 84, 84) frames whose bytes are a function of their global row; samples drawn from the episodes past the 2^31- and 2^32-byte
    marks (and from the first, for contrast) must come back bit-exact, n-step frame included."""
    from exorl_amd.engine import ReplayEngine
    from exorl_amd import _lib as L
    OB, A, T, E, nstep = 3 * 84 * 84, 2, 250, 880, 3
    rows = E * (T + 1)
    assert rows * OB > (1 << 32) + (1 << 28)
    eng = ReplayEngine((3, 84, 84), np.uint8, A, 0, rows + 8, E + 4)
    base = (np.arange(OB) % 251).astype(np.uint8)

    def frames(r0, n):       # frame of global row r: base + (37 r + (r >> 8)) mod 256, cheap to regenerate for any row
        off = ((37 * np.arange(r0, r0 + n) + (np.arange(r0, r0 + n) >> 8)) & 255).astype(np.uint8)
        return (base[None, :] + off[:, None]).reshape(n, 3, 84, 84)
    slots = []
    for e in range(E):
        r0 = e * (T + 1)
        act = np.tile(np.float32([e, 0.5]), (T + 1, 1))
        slots.append(eng.append_episode(dict(observation=frames(r0, T + 1), action=act, reward=np.full((T + 1, 1), 1.0, np.float32),
                                             discount=np.ones((T + 1, 1), np.float32))))
    eng.set_order(slots)
    picks = [0, 1, E // 2, (1 << 31) // (OB * (T + 1)) + 1, (1 << 32) // (OB * (T + 1)) + 1, E - 2, E - 1]
    pairs = np.array([[e, idx] for e in picks for idx in (1, 17, T - nstep + 1)], np.int32)
    obs, act, rew, disc, nobs = eng.sample(len(pairs), nstep, 0.99, L.SAMPLER_GIVEN, pairs=pairs)
    for b, (e, idx) in enumerate(pairs):
        r = e * (T + 1) + idx
        assert np.array_equal(obs[b].cpu().numpy(), frames(r - 1, 1)[0]), (e, idx)
        assert np.array_equal(nobs[b].cpu().numpy(), frames(r + nstep - 1, 1)[0]), (e, idx)
        assert float(act[b, 0]) == e
    assert (pairs[:, 0].astype(np.int64) * (T + 1) * OB).max() > (1 << 32)
