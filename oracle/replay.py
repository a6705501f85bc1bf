"""ORACLE (test infrastructure only — never imported by the product path).

CPU restatement of the reference's episodic replay sampler,
/root/reference/utils/replay_buffer.py:153-239 (ReplayBuffer) and :45-100 (OfflineReplayBuffer).
Pinned bit-exactly by tests/golden/replay_*.npz.
"""
import numpy as np

from .mt19937 import MT19937


def episode_len(ep):
    # replay_buffer.py:13-15 — first row is the dummy reset transition
    return next(iter(ep.values())).shape[0] - 1


def episode_name(idx, length):
    # replay_buffer.py:149
    return f'episode_{idx}_{length}.npz'


class OracleReplay:
    """State machine of ReplayBuffer (_try_fetch/_store_episode/_sample), directory = dict name->episode."""

    def __init__(self, max_size, num_workers, nstep, discount, fetch_every=1000, worker_id=0,
                 meta_keys=()):
        self.max_size = max_size                     # replay_buffer.py:262 (already // workers)
        self.num_workers = max(1, num_workers)       # :159
        self.worker_id = worker_id
        self.nstep = nstep
        self.discount = discount
        self.fetch_every = fetch_every
        self.since_fetch = fetch_every               # :165
        self.size = 0
        self.fns = []                                # :160, kept sorted (:184)
        self.episodes = {}
        self.meta_keys = tuple(meta_keys)
        self.py_rng = None                           # random.*  stream  (:169)
        self.np_rng = None                           # np.random.* stream (:222)

    def seed(self, py_seed, np_seed):
        self.py_rng = MT19937.python_seed(py_seed)
        self.np_rng = MT19937.numpy_seed(np_seed)

    # -- :172-190
    def _store_episode(self, name, ep, directory):
        L = episode_len(ep)
        while L + self.size > self.max_size:
            early = self.fns.pop(0)                  # lexicographically first, not oldest
            self.size -= episode_len(self.episodes.pop(early))
            directory.pop(early, None)               # unlink(missing_ok=True)
        self.fns.append(name)
        self.fns.sort()
        self.episodes[name] = ep
        self.size += L
        return True

    # -- :192-212
    def try_fetch(self, directory):
        if self.since_fetch < self.fetch_every:
            return
        self.since_fetch = 0
        fetched = 0
        for name in sorted(directory.keys(), reverse=True):
            stem = name[:-4]
            idx, L = [int(x) for x in stem.split('_')[1:]]
            if idx % self.num_workers != self.worker_id:
                continue
            if name in self.episodes:
                break
            if fetched + L > self.max_size:
                break
            fetched += L
            self._store_episode(name, directory[name], directory)

    # -- :214-235, one sample: returns (episode name, idx)
    def draw(self, directory):
        self.try_fetch(directory)
        self.since_fetch += 1
        name = self.fns[self.py_rng.py_randbelow(len(self.fns))]
        L = episode_len(self.episodes[name])
        idx = self.np_rng.np_randint0(L - self.nstep + 1) + 1
        return name, idx

    def gather(self, name, idx):
        return gather_nstep(self.episodes[name], idx, self.nstep, self.discount, self.meta_keys)

    def sample_batch(self, directory, batch):
        draws = [self.draw(directory) for _ in range(batch)]
        cols = [self.gather(n, i) for n, i in draws]
        return draws, tuple(np.stack([c[j] for c in cols]) for j in range(len(cols[0])))


def relabel_episode(env, ep):
    """replay_buffer.py:31-42: reward[i] = task.get_reward(physics at stored state i), cast to the env's reward spec."""
    spec = env.reward_spec()
    rewards = []
    for st in ep['physics']:
        with env.physics.reset_context():
            env.physics.set_state(st)
        rewards.append(np.full(spec.shape, env.task.get_reward(env.physics), spec.dtype))
    out = dict(ep)
    out['reward'] = np.array(rewards, dtype=spec.dtype)
    return out


class OracleOfflineReplay:
    """OfflineReplayBuffer (replay_buffer.py:45-100): one ascending scan of the directory at the first sample, stop once the running
    size EXCEEDS max_size (:62-63, so the last episode loaded overshoots), worker modulo (:65-66), optional reward relabelling,
    nothing evicted or deleted; samples are nstep=1 (:84-96). The shipped class cannot get past `_load` (it calls
    `_relable_reward`, the method is `_relabel_reward`); this restates the evident intent, pinned by replay_offline_*.npz."""

    def __init__(self, env, max_size, num_workers, discount, worker_id=0, relabel=True):
        self.env, self.max_size, self.num_workers, self.discount = env, max_size, max(1, num_workers), discount
        self.worker_id, self.relabel = worker_id, relabel
        self.size, self.fns, self.episodes, self.loaded = 0, [], {}, False
        self.py_rng = self.np_rng = None

    def seed(self, py_seed, np_seed):
        self.py_rng = MT19937.python_seed(py_seed)
        self.np_rng = MT19937.numpy_seed(np_seed)

    def load(self, directory):
        for name in sorted(directory.keys()):
            if self.size > self.max_size:
                break
            idx, _ = [int(x) for x in name[:-4].split('_')[1:]]
            if idx % self.num_workers != self.worker_id:
                continue
            ep = directory[name]
            if self.relabel:
                ep = relabel_episode(self.env, ep)
            self.fns.append(name)
            self.episodes[name] = ep
            self.size += episode_len(ep)
        self.loaded = True

    def draw(self, directory):
        if not self.loaded:
            self.load(directory)
        name = self.fns[self.py_rng.py_randbelow(len(self.fns))]
        idx = self.np_rng.np_randint0(episode_len(self.episodes[name])) + 1
        return name, idx

    def sample_batch(self, directory, batch):
        draws = [self.draw(directory) for _ in range(batch)]
        cols = [gather_nstep(self.episodes[n], i, 1, self.discount) for n, i in draws]
        return draws, tuple(np.stack([c[j] for c in cols]) for j in range(5))


def gather_nstep(ep, idx, nstep, gamma, meta_keys=()):
    """replay_buffer.py:223-235. fp32, products and sums rounded separately (no FMA)."""
    obs = ep['observation'][idx - 1]
    action = ep['action'][idx]
    next_obs = ep['observation'][idx + nstep - 1]
    reward = np.zeros_like(ep['reward'][idx])
    discount = np.ones_like(ep['discount'][idx])
    g32 = np.float32(gamma)
    for i in range(nstep):
        step_reward = ep['reward'][idx + i]
        reward = (reward + (discount * step_reward).astype(np.float32)).astype(np.float32)
        discount = (discount * (ep['discount'][idx + i] * g32).astype(np.float32)).astype(np.float32)
    meta = tuple(ep[k][idx - 1] for k in meta_keys)
    return (obs, action, reward, discount, next_obs) + meta


def gather_nstep_batch(obs, act, rew, disc, row0, idx, nstep, gamma, meta=None):
    """Vectorised form over a flat row arena (the layout the HIP path uses).

    obs/act/rew/disc: (rows, .) arrays holding all episodes back to back; row0[b] = arena row of
    the sampled episode's dummy first row; idx[b] = start index within the episode (>= 1).
    """
    r = row0 + idx
    o = obs[r - 1]
    a = act[r]
    no = obs[r + nstep - 1]
    R = np.zeros((len(r), 1), np.float32)
    D = np.ones((len(r), 1), np.float32)
    g32 = np.float32(gamma)
    for i in range(nstep):
        R = (R + (D * rew[r + i]).astype(np.float32)).astype(np.float32)
        D = (D * (disc[r + i] * g32).astype(np.float32)).astype(np.float32)
    out = (o, a, R, D, no)
    if meta is not None:
        out = out + (meta[r - 1],)
    return out


# ---------------------------------------------------------------------------------------------
# Counter-based sampler of the HIP "philox" mode (the build's own spec; distribution-equivalent
# to the reference: episode ~ U{0..E-1}, start ~ U{1..len-n+1}; SURVEY.md A1 last bullet).
# ---------------------------------------------------------------------------------------------
PHILOX_M0, PHILOX_M1 = 0xD2511F53, 0xCD9E8D57
PHILOX_W0, PHILOX_W1 = 0x9E3779B9, 0xBB67AE85


def philox4x32_10(ctr, key):
    c = [int(x) for x in ctr]
    k0, k1 = int(key[0]), int(key[1])
    for _ in range(10):
        p0 = PHILOX_M0 * c[0]
        p1 = PHILOX_M1 * c[2]
        c = [((p1 >> 32) ^ c[1] ^ k0) & 0xFFFFFFFF, p1 & 0xFFFFFFFF,
             ((p0 >> 32) ^ c[3] ^ k1) & 0xFFFFFFFF, p0 & 0xFFFFFFFF]
        k0 = (k0 + PHILOX_W0) & 0xFFFFFFFF
        k1 = (k1 + PHILOX_W1) & 0xFFFFFFFF
    return c


def philox_draw(seed, batch_counter, sample, n_episodes, ep_len, nstep):
    """Index pair for sample `sample` of batch `batch_counter`:
    x = philox(ctr=(sample, batch_lo, batch_hi, 0), key=(seed_lo, seed_hi));
    episode = mulhi(x0, E); start = mulhi(x1, len-n+1) + 1   (Lemire multiply-shift, no rejection)."""
    x = philox4x32_10((sample, batch_counter & 0xFFFFFFFF, batch_counter >> 32, 0),
                      (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF))
    e = (x[0] * n_episodes) >> 32
    span = int(ep_len[e]) - nstep + 1
    s = ((x[1] * span) >> 32) + 1
    return e, s
