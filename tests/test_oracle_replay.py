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
