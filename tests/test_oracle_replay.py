"""Oracle vs the reference's own outputs (tests/golden/replay_*.npz, made by tools/gen_golden.py).
Pins replay_buffer.py:153-239 bit-exactly: episode ordering, eviction, both MT19937 streams,
n-step reward/discount arithmetic."""
import random

import numpy as np
import pytest

import _synth
from oracle.mt19937 import MT19937
from oracle.replay import OracleReplay, episode_name, gather_nstep_batch

SCEN = ['a_nstep1', 'b_nstep3', 'c_evict', 'd_meta', 'e_single', 'f_pixels']


def test_mt19937_matches_live_generators():
    for seed in (0, 1, 3, 12345, 2**31 + 7, 2**40 + 3):
        random.seed(seed)
        g = MT19937.python_seed(seed)
        for n in (1, 2, 3, 13, 1000, 2**20 + 1):
            for _ in range(20):
                assert g.py_randbelow(n) == random.randrange(n)
    for seed in (0, 1, 3, 12345, 2**32 - 1):
        np.random.seed(seed)
        g = MT19937.numpy_seed(seed)
        for hi in (1, 2, 3, 13, 999, 1000, 2**20 + 1):
            for _ in range(20):
                assert g.np_randint0(hi) == np.random.randint(0, hi)
    # from_state hand-off (what the loader does at iter() time)
    random.seed(99)
    random.random()
    st = random.getstate()[1]
    g = MT19937.from_state(st[:-1], st[-1])
    assert [g.py_randbelow(77) for _ in range(50)] == [random.randrange(77) for _ in range(50)]


@pytest.mark.parametrize('name', SCEN)
def test_replay_golden(gold, name):
    z = np.load(gold / f'replay_{name}.npz')
    O, A, M, nstep, max_size, B, NB, seed, u8 = [int(x) for x in z['dims']]
    lengths = [int(x) for x in z['lengths']]
    eps = _synth.synth_episodes(seed, lengths, O, A, M, bool(u8))
    directory = {episode_name(i, L): ep for i, (L, ep) in enumerate(zip(lengths, eps))}
    rb = OracleReplay(max_size, 0, nstep, 0.99, meta_keys=('skill',) if M else ())
    rb.seed(seed, seed)
    picks, starts = [], []
    for bi in range(NB):
        draws, batch = rb.sample_batch(directory, B)
        picks += [int(n.split('_')[1]) for n, _ in draws]
        starts += [i for _, i in draws]
        for ti, t in enumerate(batch):
            ref = z[f'batch{bi}_{ti}']
            assert t.dtype == ref.dtype and t.shape == ref.shape
            assert np.array_equal(t.view(np.uint8), ref.view(np.uint8)), (name, bi, ti)
    assert picks == list(z['picks']) and starts == list(z['starts'])
    assert [int(n.split('_')[1]) for n in rb.fns] == list(z['resident'])


class FakeEnv:
    """The stand-in tools/gen_golden.py handed to the reference's relabel_episode (replay_buffer.py:31-42)."""

    class _Physics:
        def reset_context(self):
            import contextlib
            return contextlib.nullcontext()

        def set_state(self, s):
            self.state = np.array(s, np.float64)

    class _Task:
        def get_reward(self, physics):
            return float(np.tanh(physics.state).sum() * 0.25)

    class _Spec:
        shape, dtype = (1,), np.dtype(np.float32)

    def __init__(self):
        self.physics, self.task = self._Physics(), self._Task()

    def reward_spec(self):
        return self._Spec()


def offline_episodes(z):
    O, A, P, max_size, B, NB, seed, relabel = [int(x) for x in z['dims']]
    lengths = [int(x) for x in z['lengths']]
    eps = _synth.synth_episodes(seed, lengths, O, A)
    rs = np.random.RandomState(100 + seed)
    for ep in eps:
        ep['physics'] = rs.standard_normal((ep['observation'].shape[0], P))
    return eps, lengths


OFFLINE = ['offline_a', 'offline_b_cap', 'offline_c_relabel']


@pytest.mark.parametrize('name', OFFLINE)
def test_offline_replay_golden(gold, name):
    """OfflineReplayBuffer semantics (replay_buffer.py:45-100) against the reference's own batches: ascending load with the
    `size > max_size` stop, nstep=1 samples, relabelled rewards."""
    from oracle.replay import OracleOfflineReplay
    z = np.load(gold / f'replay_{name}.npz')
    O, A, P, max_size, B, NB, seed, relabel = [int(x) for x in z['dims']]
    eps, lengths = offline_episodes(z)
    directory = {episode_name(i, L): ep for i, (L, ep) in enumerate(zip(lengths, eps))}
    rb = OracleOfflineReplay(FakeEnv(), max_size, 0, 0.99, relabel=bool(relabel))
    rb.seed(seed, seed)
    for bi in range(NB):
        _, batch = rb.sample_batch(directory, B)
        for ti, t in enumerate(batch):
            ref = z[f'batch{bi}_{ti}']
            assert t.dtype == ref.dtype and t.shape == ref.shape, (t.dtype, ref.dtype, t.shape, ref.shape)
            assert np.array_equal(t.view(np.uint8), ref.view(np.uint8)), (name, bi, ti)
    assert [int(n.split('_')[1]) for n in rb.fns] == list(z['resident']) and rb.size == int(z['size'])
    if relabel:
        for i in z['resident']:
            assert np.array_equal(rb.episodes[episode_name(int(i), lengths[int(i)])]['reward'], z[f'reward{int(i)}'])


def test_flat_arena_gather_equals_episode_gather():
    lengths = [7, 3, 12, 5]
    eps = _synth.synth_episodes(1, lengths, 5, 2, 2)
    cat = lambda k: np.concatenate([e[k] for e in eps])
    row0 = np.cumsum([0] + [L + 1 for L in lengths[:-1]])
    rs = np.random.RandomState(0)
    e = rs.randint(0, 4, 64)
    idx = np.array([rs.randint(0, lengths[i] - 3 + 1) + 1 for i in e])
    out = gather_nstep_batch(cat('observation'), cat('action'), cat('reward'), cat('discount'), row0[e], idx, 3, 0.99,
                             cat('skill'))
    from oracle.replay import gather_nstep
    for b in range(64):
        one = gather_nstep(eps[e[b]], idx[b], 3, 0.99, ('skill',))
        for x, y in zip(out, one):
            assert np.array_equal(x[b], y)
