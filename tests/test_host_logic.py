"""CPU-only checks of the product's host side: the C-ABI library loads and exports every symbol the header
declares, and the Python mirror's init plumbing reproduces the reference's initial weights."""
import ctypes
import re
from pathlib import Path

import numpy as np
import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent


def _declared_symbols():
    text = (ROOT / 'include' / 'exorl_hip.h').read_text()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(exorl_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    from exorl_amd import build, _lib
    build.build(verbose=False)
    lib = ctypes.CDLL(str(_lib.LIB_PATH))
    syms = _declared_symbols()
    assert len(syms) >= 35
    for s in syms:
        assert hasattr(lib, s), f'{s} declared in include/exorl_hip.h but not exported'
        assert s in _lib.PROTOTYPES, f'{s} has no ctypes prototype'
    assert set(_lib.PROTOTYPES) == set(syms)
    lib.exorl_abi_version.restype = ctypes.c_int
    version = int(re.search(r'#define EXORL_ABI_VERSION (\d+)', (ROOT / 'include' / 'exorl_hip.h').read_text()).group(1))
    assert lib.exorl_abi_version() == version


def test_product_fails_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from exorl_amd import agents, _lib
    with pytest.raises(_lib.ExorlError):
        agents.TD3BCAgent('td3_bc', (5,), (3,), 'cuda', 1e-4, 32, 0.01, 0.2, 1, 8, 0.3, True, 2.5)
    with pytest.raises(_lib.ExorlError):
        agents.TD3BCAgent('td3_bc', (5,), (3,), 'cpu', 1e-4, 32, 0.01, 0.2, 1, 8, 0.3, True, 2.5)


@pytest.mark.parametrize('kind', ['td3_bc', 'bc', 'ddpg'])
def test_init_matches_reference_rng_order(gold, kind):
    """torch.manual_seed(21) + the reference constructor (fixture) == our _mlp_init draw order."""
    from exorl_amd import agents
    z = np.load(gold / f'tiny_{kind}.npz')
    O, A, H = 5, 3, 32
    torch.manual_seed(21)
    actor0 = agents._mlp_init(O, H, A, 1, 1)
    keys = agents._DDPG_ACTOR_KEYS if kind == 'ddpg' else agents._OFFLINE_ACTOR_KEYS
    for k, w in zip(keys, actor0):
        np.testing.assert_array_equal(w.numpy(), z[f'init/actor/{k}'], err_msg=k)
    if kind != 'bc':
        critic0 = agents._mlp_init(O + A, H, 1, 1 if kind == 'ddpg' else 2, 2)
        ckeys = agents._DDPG_CRITIC_KEYS if kind == 'ddpg' else agents._OFFLINE_CRITIC_KEYS
        for k, w in zip(ckeys, critic0):
            np.testing.assert_array_equal(w.numpy(), z[f'init/critic/{k}'], err_msg=k)


@pytest.mark.parametrize('kind', ['rnd', 'icm', 'icm_apt'])
def test_intr_module_init_matches_reference_rng_order(gold, kind):
    """The intrinsic-reward modules are constructed after the DDPG nets (rnd.py:64-71, icm.py:49-55, icm_apt.py:61-70)."""
    from exorl_amd import agents
    z = np.load(gold / f'tiny_{kind}.npz')
    O, A, H, R = 5, 3, 32, 16
    torch.manual_seed(21)
    agents._mlp_init(O, H, A, 1, 1)
    agents._mlp_init(O + A, H, 1, 1, 2)
    agents._mlp_init(O + A, H, 1, 1, 2)
    if kind == 'rnd':
        w, keys, mod = agents._seq_init([('lin', O, H), ('lin', H, H), ('lin', H, R)] * 2), agents._RND_KEYS, 'rnd'
    elif kind == 'icm':
        w, keys, mod = agents._seq_init([('lin', O + A, H), ('lin', H, O), ('lin', 2 * O, H), ('lin', H, A)]), agents._ICM_KEYS, 'icm'
    else:
        w = agents._seq_init([('lin', O, R), ('ln', R), ('lin', R + A, H), ('lin', H, R), ('lin', 2 * R, H), ('lin', H, A)])
        keys, mod = agents._APT_KEYS, 'icm'
    assert len(w) == len(keys)
    for k, t in zip(keys, w):
        np.testing.assert_array_equal(t.numpy(), z[f'init/{mod}/{k}'], err_msg=k)


def test_aps_init_matches_reference_rng_order(gold):
    """APSAgent.__init__ (aps.py:82-119): DDPG nets (scalar critics, then discarded), CriticSF x2, then the APS module."""
    from exorl_amd import agents
    z = np.load(gold / 'tiny_aps.npz')
    O, A, H, S = 5, 3, 32, 4
    torch.manual_seed(21)
    actor0 = agents._mlp_init(O + S, H, A, 1, 1)
    for _ in range(2):
        agents._mlp_init(O + S + A, H, 1, 1, 2)
    critic0 = agents._mlp_init(O + S + A, H, S, 1, 2)
    agents._mlp_init(O + S + A, H, S, 1, 2)
    w = agents._seq_init([('lin', O, H), ('lin', H, H), ('lin', H, S)])
    for k, t in zip(agents._DDPG_ACTOR_KEYS, actor0):
        np.testing.assert_array_equal(t.numpy(), z[f'init/actor/{k}'], err_msg=k)
    for k, t in zip(agents._DDPG_CRITIC_KEYS, critic0):
        np.testing.assert_array_equal(t.numpy(), z[f'init/critic/{k}'], err_msg=k)
    for k, t in zip(agents._APS_KEYS, w):
        np.testing.assert_array_equal(t.numpy(), z[f'init/aps/{k}'], err_msg=k)


def test_pixel_init_matches_reference_rng_order(gold):
    """DDPGAgent(obs_type='pixels') under torch.manual_seed(5): Encoder (orthogonal with the ReLU gain), pixel Actor, pixel Critic."""
    from exorl_amd import agents
    z = np.load(gold / 'pixel_ddpg.npz')
    C_, HW, A, F, H = [int(v) for v in z['dims'][:5]]
    torch.manual_seed(5)
    w = agents._pixel_init(C_, HW, A, F, H)
    for nm in ('encoder', 'actor', 'critic'):
        got = np.array([[float(t.double().sum()), float((t.double() ** 2).sum())] for t in w[nm]])
        np.testing.assert_allclose(got, z[f'init_sums/{nm}'], rtol=1e-6, atol=1e-6, err_msg=nm)


def test_proto_init_matches_reference_rng_order(gold):
    """proto.py:55-67 draws: predictor, projector (weight_init applied twice), protos — after the DDPG nets."""
    from exorl_amd import agents
    z = np.load(gold / 'tiny_proto.npz')
    O, A, H = 5, 3, 32
    torch.manual_seed(21)
    agents._mlp_init(O, H, A, 1, 1)
    agents._mlp_init(O + A, H, 1, 1, 2)
    agents._mlp_init(O + A, H, 1, 1, 2)
    w = agents._proto_init(O, 8, 16, 6)
    keys = ['predictor/weight', 'predictor/bias', 'projector/trunk.0.weight', 'projector/trunk.0.bias', 'projector/trunk.2.weight',
            'projector/trunk.2.bias', 'protos/weight']
    for k, t in zip(keys, w):
        np.testing.assert_array_equal(t.numpy(), z[f'init/{k}'], err_msg=k)


def test_schedule_and_helpers(gold):
    from exorl_amd import utils
    z = np.load(gold / 'utils_g2.npz')
    sch = ['0.2', 'linear(1.0,0.1,100)', 'step_linear(1.0,0.5,50,0.1,100)']
    steps = [0, 10, 50, 75, 100, 1000]
    got = np.array([[utils.schedule(s, t) for t in steps] for s in sch])
    assert np.array_equal(got, z['schedule'])
    with pytest.raises(NotImplementedError):
        utils.schedule('cosine(1,2)', 0)


def test_storage_writes_reference_format(tmp_path):
    """ReplayBufferStorage: episode_{idx}_{len}.npz, len = rows-1, arrays per spec (replay_buffer.py:115-150)."""
    from exorl_amd.replay_buffer import ReplayBufferStorage, load_episode, episode_len

    class Spec:
        def __init__(self, shape, dtype, name):
            self.shape, self.dtype, self.name = shape, np.dtype(dtype), name

    class TS(dict):
        def __init__(self, last, **kw):
            super().__init__(**kw)
            self._last = last

        def last(self):
            return self._last

    specs = (Spec((3,), np.float32, 'observation'), Spec((2,), np.float32, 'action'), Spec((1,), np.float32, 'reward'),
             Spec((1,), np.float32, 'discount'))
    st = ReplayBufferStorage(specs, (Spec((2,), np.float32, 'skill'),), tmp_path / 'buffer')
    for ep in range(2):
        for t in range(4 + ep):
            st.add(TS(t == 3 + ep, observation=np.full(3, t, np.float32), action=np.zeros(2, np.float32), reward=0.5,
                      discount=1.0), {'skill': np.ones(2, np.float32)})
    assert len(st) == 3 + 4
    names = sorted(p.name for p in (tmp_path / 'buffer').glob('*.npz'))
    assert names == ['episode_0_3.npz', 'episode_1_4.npz']
    ep = load_episode(tmp_path / 'buffer' / 'episode_1_4.npz')
    assert episode_len(ep) == 4 and ep['reward'].shape == (5, 1) and ep['reward'].dtype == np.float32
    assert ep['skill'].shape == (5, 2)
    st2 = ReplayBufferStorage(specs, (), tmp_path / 'buffer')      # _preload resumes the counters
    assert len(st2) == 7 and st2._num_episodes == 2


def test_parallel_episode_decode_keeps_order_and_stops_at_broken_file(tmp_path):
    """The ingest pool must hand episodes back in the requested order and yield None for an unreadable
    file, which is where the fetch stops (replay_buffer.py:209-212 breaks out of its loop the same way)."""
    from exorl_amd import replay_buffer as rb
    fns = []
    for i in range(12):
        ep = dict(observation=np.full((5 + i, 3), i, np.float32), action=np.zeros((5 + i, 2), np.float32),
                  reward=np.zeros((5 + i, 1), np.float32), discount=np.ones((5 + i, 1), np.float32))
        fn = tmp_path / f'episode_{i}_{4 + i}.npz'
        rb.save_episode(ep, fn)
        fns.append(fn)
    fns[7].write_bytes(b'not a zip file')
    for threads in (1, 4):
        got = list(rb._load_many(fns, threads))
        assert [None if e is None else int(e['observation'][0, 0]) for e in got] == [0, 1, 2, 3, 4, 5, 6, None, 8, 9, 10, 11]
        assert all(rb.episode_len(e) == 4 + i for i, e in enumerate(got) if e is not None)


def test_offline_file_selection_rule(tmp_path):
    """OfflineReplayBuffer._load's selection (replay_buffer.py:58-75) from file names alone: ascending lexicographic order, stop once
    the running size EXCEEDS max_size, per-worker modulo. A dataset larger than max_size keeps its FIRST episodes (the online buffer's
    reverse scan would keep the last ones)."""
    from exorl_amd.replay_buffer import _OfflineShard
    lengths = [7, 3, 12, 5, 9, 4, 3, 15, 6, 8, 10, 3, 11]
    for i, n in enumerate(lengths):
        (tmp_path / f'episode_{i}_{n}.npz').write_bytes(b'')
    idx = lambda fns: [int(f.stem.split('_')[1]) for f in fns]
    todo, size = _OfflineShard.select(tmp_path, 40, 1, 0)
    assert idx(todo) == [0, 10, 11, 12, 1, 2] and size == 46            # == tests/golden/replay_offline_b_cap.npz 'resident'
    todo, size = _OfflineShard.select(tmp_path, 10**6, 1, 0)
    assert idx(todo) == [0, 10, 11, 12, 1, 2, 3, 4, 5, 6, 7, 8, 9] and size == sum(lengths)
    todo, size = _OfflineShard.select(tmp_path, 20, 2, 1)                 # worker 1 of 2: odd episode indices only
    assert idx(todo) == [11, 1, 3, 5, 7] and size == 3 + 3 + 5 + 4 + 15   # 15 <= 20 after four files, so a fifth is taken


def test_until_every_timer_semantics():
    """utils.Until / Every / Timer as the training loops use them (reference utils.py:87-125, pretrain.py:209-215): frame counts are divided by
    action_repeat at call time, None switches the gate off."""
    import time
    from exorl_amd import utils
    u = utils.Until(100, action_repeat=2)
    assert [u(s) for s in (0, 49, 50, 51)] == [True, True, False, False]
    assert utils.Until(None)(10**9) is True
    e = utils.Every(10, action_repeat=2)
    assert [e(s) for s in (0, 4, 5, 10, 11)] == [True, False, True, True, False]
    assert utils.Every(None)(0) is False
    assert utils.Until(7)(6) and not utils.Until(7)(7) and utils.Every(3)(9) and not utils.Every(3)(10)
    t = utils.Timer()
    time.sleep(0.01)
    lap, total = t.reset()
    assert 0.005 < lap <= total + 1e-9
    lap2, total2 = t.reset()
    assert lap2 < lap and total2 >= total and t.total_time() >= total2
