"""Host-side mirror of the reference's replay interface, backed by the HBM arena in libexorl_hip.so.

Same names, arguments and on-disk format as /root/reference/utils/replay_buffer.py:
  ReplayBufferStorage(data_specs, meta_specs, replay_dir)      :103-150   (episode_{idx}_{len}.npz writer)
  make_replay_loader(storage, max_size, batch_size, num_workers, save_snapshot, nstep, discount)  :260-277
  iter(loader) / next(it) -> (obs, action, reward, discount, next_obs, *meta) batched           :214-239

What is host logic here (file discovery, lexicographic ordering, eviction bookkeeping — :172-212) stays in
Python; what was per-sample Python (episode pick, start index, gathers, n-step loop — :214-235, 30 us/sample)
is one HIP launch per batch. Tensors come back on the GPU (the reference's next step was an H2D copy,
utils.py:55-56, so agents accept either).
"""
import bisect
import os
from collections import OrderedDict, defaultdict
from pathlib import Path

import numpy as np
import torch

from . import _lib as L
from .engine import ReplayEngine


def episode_len(episode):
    """Transitions in an episode dict: every array carries one extra leading row, the reset step (replay_buffer.py:13-15)."""
    first = next(iter(episode.values()))
    return len(first) - 1


def save_episode(episode, fn):
    """One episode -> one compressed .npz (the reference's on-disk format, replay_buffer.py:18-23). Written under a temporary name and
    renamed, so a loader scanning the directory never sees a half-written file under a name its glob('*.npz') matches."""
    fn = Path(fn)
    tmp = fn.with_name(fn.name + '.part')
    with tmp.open('wb') as f:
        np.savez_compressed(f, **episode)
    os.replace(tmp, fn)


def load_episode(fn):
    """{key: array} of one episode file (replay_buffer.py:25-29); arrays are materialised before the file is closed."""
    with np.load(Path(fn)) as z:
        return {k: z[k] for k in z.files}


def _load_or_none(fn):
    try:
        return load_episode(fn)
    except Exception:            # a file still being written: the reference stops the fetch there too (replay_buffer.py:209-212)
        return None


def _load_many(fns, threads):
    """Decodes episode files in order with a bounded look-ahead (a 5-10 M-transition dataset is thousands of
    zlib-compressed .npz files; one decoding thread is what made loading minutes)."""
    if threads <= 1 or len(fns) < 2:
        for fn in fns:
            yield _load_or_none(fn)
        return
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=threads) as pool:
        window = 4 * threads
        pending = [pool.submit(_load_or_none, fn) for fn in fns[:window]]
        nxt = len(pending)
        for i in range(len(fns)):
            ep = pending[i].result()
            pending[i] = None
            if nxt < len(fns):
                pending.append(pool.submit(_load_or_none, fns[nxt]))
                nxt += 1
            yield ep


class ReplayBufferStorage:
    """Accumulates time-steps and writes finished episodes as episode_{idx}_{len}.npz."""

    def __init__(self, data_specs, meta_specs, replay_dir):
        self._data_specs = data_specs
        self._meta_specs = meta_specs
        self._replay_dir = Path(replay_dir)
        self._replay_dir.mkdir(exist_ok=True)
        self._current_episode = defaultdict(list)
        self._num_episodes = 0
        self._num_transitions = 0
        # episodes finished in this process, kept until an in-process loader has taken them: the HBM sampler ingests them from here
        # instead of re-reading and inflating the file it was just written to; bounded, oldest dropped first
        self._fresh = OrderedDict()
        self._fresh_max = 64
        for fn in self._replay_dir.glob('*.npz'):           # resume counters from what is on disk
            self._num_episodes += 1
            self._num_transitions += int(fn.stem.split('_')[2])

    def __len__(self):
        return self._num_transitions

    def add(self, time_step, meta):
        """One environment step (replay_buffer.py:120-141): meta values and the spec'd fields of `time_step` are appended to the open
        episode; scalars are broadcast to their spec's shape; the step that ends the episode flushes it to disk."""
        cur = self._current_episode
        for key, value in meta.items():
            cur[key].append(value)
        for spec in self._data_specs:
            value = time_step[spec.name]
            if np.isscalar(value):
                value = np.full(spec.shape, value, spec.dtype)
            if value.shape != spec.shape or value.dtype != spec.dtype:
                raise AssertionError(f'{spec.name}: got {value.shape} {value.dtype}, spec says {spec.shape} {spec.dtype}')
            cur[spec.name].append(value)
        if not time_step.last():
            return
        self._current_episode = defaultdict(list)
        self._store_episode({spec.name: np.array(cur[spec.name], spec.dtype) for spec in (*self._data_specs, *self._meta_specs)})

    def _store_episode(self, episode):
        idx, length = self._num_episodes, episode_len(episode)
        self._num_episodes += 1
        self._num_transitions += length
        fn = self._replay_dir / f'episode_{idx}_{length}.npz'
        save_episode(episode, fn)
        self._fresh[fn] = episode
        while len(self._fresh) > self._fresh_max:
            self._fresh.popitem(last=False)


class _Shard:
    """One reference 'worker': its own resident set, limits and index streams (replay_buffer.py:153-212)."""

    def __init__(self, loader, worker_id):
        self.loader = loader
        self.worker_id = worker_id
        self.size = 0
        self.fns = []                  # sorted, like ReplayBuffer._episode_fns
        self.slot = {}                 # fn -> arena slot
        self.length = {}
        self.since_fetch = loader.fetch_every
        self.dir_stamp = None
        self.engine = None
        self.seeded = False

    def _ensure_engine(self, episode):
        if self.engine is not None:
            return
        ld = self.loader
        obs = episode['observation']
        meta_dim = sum(int(np.prod(episode[k].shape[1:])) for k in ld.meta_keys)
        max_eps = ld.max_episodes or max(4096, ld.max_size // 64)
        cap = ld.capacity_rows or (ld.max_size + max_eps + obs.shape[0] + 1024)
        self.engine = ReplayEngine(obs.shape[1:], obs.dtype, int(np.prod(episode['action'].shape[1:])), meta_dim, cap,
                                   max_eps, ld.device)

    def _store(self, fn, episode=None, reorder=True):
        ld = self.loader
        if episode is None:
            try:
                episode = load_episode(fn)
            except Exception:
                return False
        n = episode_len(episode)
        self._ensure_engine(episode)
        while n + self.size > ld.max_size:
            early = self.fns.pop(0)                       # lexicographically first (replay_buffer.py:178-182)
            self.engine.evict(self.slot.pop(early))
            self.size -= self.length.pop(early)
            early.unlink(missing_ok=True)
        self.slot[fn] = self.engine.append_episode(episode, ld.meta_keys)
        self.length[fn] = n
        bisect.insort(self.fns, fn)                       # == append + sort (replay_buffer.py:185-186), without re-sorting per episode
        self.size += n
        if reorder:
            self.engine.set_order([self.slot[f] for f in self.fns])
        if not ld.save_snapshot:
            fn.unlink(missing_ok=True)
        return True

    def try_fetch(self):
        ld = self.loader
        if self.since_fetch < ld.fetch_every:
            return
        self.since_fetch = 0
        d = ld.storage._replay_dir
        stamp = os.stat(d).st_mtime_ns
        if stamp == self.dir_stamp and self.fns:
            return                                         # nothing was added since the last scan
        fetched, todo = 0, []
        for fn in sorted(d.glob('*.npz'), reverse=True):  # the reference's selection (replay_buffer.py:196-212) ...
            idx, n = (int(x) for x in fn.stem.split('_')[1:])
            if idx % ld.num_workers != self.worker_id:
                continue
            if fn in self.slot:
                break
            if fetched + n > ld.max_size:
                break
            fetched += n
            todo.append(fn)
        # ... decoded by a thread pool (zlib releases the GIL), stored in the reference's order; the episode table is uploaded once
        stored = False
        fresh = getattr(ld.storage, '_fresh', None)
        if fresh and ld.num_workers == 1:                 # one consumer: hand the in-memory copies over and let the storage forget them
            have = {fn: fresh.pop(fn) for fn in todo if fn in fresh}
            rest = iter(_load_many([fn for fn in todo if fn not in have], ld.load_threads))
            episodes = (have[fn] if fn in have else next(rest) for fn in todo)
        else:
            episodes = _load_many(todo, ld.load_threads)
        for fn, episode in zip(todo, episodes):
            if episode is None or not self._store(fn, episode, reorder=False):
                break
            stored = True
        if stored:
            self.engine.set_order([self.slot[f] for f in self.fns])
        self.dir_stamp = os.stat(d).st_mtime_ns


def relabel_episode(env, episode):
    """Rewards recomputed from the stored simulator states under `env`'s task (replay_buffer.py:31-42): ExORL datasets are
    collected reward-free and labelled for the task at load time. `env` is a dm_control-style environment (physics.reset_context /
    set_state, task.get_reward, reward_spec); there is no MuJoCo in this repo, the call is host glue around the caller's env."""
    if 'physics' not in episode:
        raise KeyError("relabel_episode: the episode has no 'physics' states to replay (pass env=None / relabel=False to keep "
                       "the stored rewards)")
    spec = env.reward_spec()
    out = np.empty((len(episode['physics']),) + tuple(spec.shape), spec.dtype)
    for i, state in enumerate(episode['physics']):
        with env.physics.reset_context():
            env.physics.set_state(state)
        out[i] = env.task.get_reward(env.physics)
    episode = dict(episode)
    episode['reward'] = out
    return episode


class _OfflineShard(_Shard):
    """One worker of OfflineReplayBuffer (replay_buffer.py:45-100): a single ASCENDING scan when the first sample is asked for,
    episodes taken until the running size exceeds max_size (:62-63 — checked before each file, so the last one overshoots),
    worker modulo (:65-66), rewards relabelled when an env is given, nothing evicted, re-scanned or deleted. The arena is sized
    from the selected files (lengths are in the names), not from max_size."""

    @staticmethod
    def select(replay_dir, max_size, num_workers, worker_id):
        """The files `_load` keeps (replay_buffer.py:58-75) and their total length, from the names alone."""
        todo, size = [], 0
        for fn in sorted(Path(replay_dir).glob('*.npz')):
            if size > max_size:
                break
            idx, n = (int(x) for x in fn.stem.split('_')[1:])
            if idx % num_workers != worker_id:
                continue
            todo.append(fn)
            size += n
        return todo, size

    def try_fetch(self):
        if self.engine is not None or self.since_fetch < 0:
            return
        ld = self.loader
        todo, size = self.select(ld.storage._replay_dir, ld.max_size, ld.num_workers, self.worker_id)
        if not todo:
            return
        ld.max_episodes, ld.capacity_rows = len(todo) + 8, size + len(todo) + 64
        for fn, episode in zip(todo, _load_many(todo, ld.load_threads)):
            if episode is None:
                raise IOError(f'offline dataset: cannot read {fn}')
            if ld.relabel:
                episode = relabel_episode(ld.env, episode)
            self._ensure_engine(episode)
            self.slot[fn] = self.engine.append_episode(episode, ld.meta_keys)
            self.length[fn] = episode_len(episode)
            self.fns.append(fn)
            self.size += self.length[fn]
        self.engine.set_order([self.slot[f] for f in self.fns])
        self.since_fetch = -(1 << 62)            # loaded once (replay_buffer.py:78-80)


class DeviceReplayLoader:
    """What make_replay_loader returns: iterable whose iterator yields device-resident minibatches."""

    def __init__(self, storage, max_size, batch_size, num_workers, save_snapshot, nstep, discount, fetch_every=1000,
                 device='cuda', sampler='mt19937', seed=None, worker_ids=None, max_episodes=None, capacity_rows=None, static=False,
                 load_threads=None, offline=False, env=None, relabel=False):
        self.storage = storage
        self.offline, self.env, self.relabel = offline, env, relabel      # OfflineReplayBuffer semantics (see _OfflineShard)
        self.static = static                # the directory will not change (offline datasets): load once, never re-scan
        self.num_workers = max(1, num_workers)
        self.max_size = max_size // self.num_workers          # replay_buffer.py:262
        self.batch_size = batch_size
        self.save_snapshot = save_snapshot
        self.nstep = nstep
        self.discount = discount
        self.fetch_every = fetch_every
        self.device = device
        self.sampler = {'mt19937': L.SAMPLER_MT19937, 'philox': L.SAMPLER_PHILOX}[sampler]
        self.seed = seed
        self.meta_keys = tuple(s.name for s in getattr(storage, '_meta_specs', ()))
        self.max_episodes = max_episodes
        self.capacity_rows = capacity_rows
        self.load_threads = load_threads if load_threads is not None else min(16, os.cpu_count() or 1)      # episode decode pool
        # which reference workers this process plays: all of them (single process) or a subset (one per DP rank)
        self.worker_ids = list(range(self.num_workers)) if worker_ids is None else list(worker_ids)

    def __iter__(self):
        return DeviceReplayIterator(self)


class DeviceReplayIterator:
    def __init__(self, loader):
        self.loader = loader
        self.shards = [(_OfflineShard if loader.offline else _Shard)(loader, w) for w in loader.worker_ids]
        self.turn = 0
        self._seeded = False
        # what agent.enable_graph reads; .engine is only set for a static single-shard dataset (see static_engine)
        self.sampler, self.nstep, self.discount, self.batch_size = loader.sampler, loader.nstep, loader.discount, loader.batch_size

    @property
    def engine(self):
        """The HBM arena, when sampling from it can be captured into the agent's hipGraph: a static directory held by one
        shard with the Philox sampler (no host-side re-scan or MT19937 draws between steps). None otherwise."""
        ld = self.loader
        if not (ld.static and len(self.shards) == 1 and ld.sampler == L.SAMPLER_PHILOX):
            return None
        shard = self.shards[0]
        if shard.engine is None:
            shard.try_fetch()
            if shard.engine is None:
                raise IndexError('replay buffer is empty (random.choice on an empty list, replay_buffer.py:169)')
            self._seed(shard)
        shard.since_fetch = -(1 << 62)      # the captured graph samples without passing through the fetch counter
        return shard.engine

    def __iter__(self):
        return self

    def _seed(self, shard):
        ld = self.loader
        if ld.sampler == L.SAMPLER_MT19937:
            if ld.num_workers == 1 and ld.seed is None:
                shard.engine.seed_mt_from_globals()          # continue random / np.random exactly (num_workers=0 case)
            else:                                            # _worker_init_fn: seed = np state word 0 + worker_id
                base = int(np.random.get_state()[1][0]) if ld.seed is None else int(ld.seed)
                shard.engine.seed_mt_ints(base + shard.worker_id, (base + shard.worker_id) & 0xFFFFFFFF)
        else:
            base = int(np.random.get_state()[1][0]) if ld.seed is None else int(ld.seed)
            shard.engine.seed_philox(base + shard.worker_id)
        shard.seeded = True

    def _segments(self, shard, batch):
        """Splits a batch where the reference would re-scan the directory mid-batch: it checks before EVERY
        sample (replay_buffer.py:216,193) and scans once `fetch_every` samples have been drawn."""
        done = 0
        while done < batch:
            if shard.since_fetch >= self.loader.fetch_every:
                shard.try_fetch()
                if not getattr(shard, 'seeded', False) and shard.engine is not None:
                    self._seed(shard)
            n = min(self.loader.fetch_every - shard.since_fetch, batch - done)
            yield done, n
            shard.since_fetch += n
            done += n

    def sample_into(self, out, batch=None):
        """Zero-copy path: writes the next minibatch straight into caller-owned device memory (exorl_batch_out)."""
        ld = self.loader
        batch = batch or ld.batch_size
        shard = self.shards[self.turn % len(self.shards)]
        self.turn += 1
        for start, n in self._segments(shard, batch):
            if shard.engine is None or not shard.fns:
                raise IndexError('replay buffer is empty (random.choice on an empty list, replay_buffer.py:169)')
            seg = L.BatchOut(out.obs + start * out.obs_stride, out.obs_stride,
                             out.action + start * out.action_stride * 4, out.action_stride,
                             out.reward + start * 4, out.discount + start * 4,
                             out.next_obs + start * out.next_obs_stride, out.next_obs_stride,
                             (out.meta + start * out.meta_stride * 4) if out.meta else None, out.meta_stride)
            shard.engine.sample_into(seg, n, ld.nstep, ld.discount, ld.sampler)

    def __next__(self):
        ld = self.loader
        B = ld.batch_size
        shard = self.shards[self.turn % len(self.shards)]
        if shard.engine is None:                              # first batch: the scan that sample 0 would trigger
            shard.try_fetch()
            if shard.engine is None:
                raise IndexError('replay buffer is empty (random.choice on an empty list, replay_buffer.py:169)')
            self._seed(shard)
        eng = shard.engine
        tdt = torch.uint8 if eng.obs_dtype == np.uint8 else torch.float32
        obs = torch.empty((B,) + eng.obs_shape, dtype=tdt, device=eng.device)
        nobs = torch.empty_like(obs)
        act = torch.empty(B, eng.act_dim, dtype=torch.float32, device=eng.device)
        rew = torch.empty(B, 1, dtype=torch.float32, device=eng.device)
        disc = torch.empty(B, 1, dtype=torch.float32, device=eng.device)
        meta = torch.empty(B, eng.meta_dim, dtype=torch.float32, device=eng.device) if eng.meta_dim else None
        out = L.BatchOut(obs.data_ptr(), eng.obs_bytes, act.data_ptr(), eng.act_dim, rew.data_ptr(), disc.data_ptr(),
                         nobs.data_ptr(), eng.obs_bytes, L.ptr(meta), eng.meta_dim)
        self.sample_into(out, B)
        res = [obs, act, rew, disc, nobs]
        if meta is not None:                                  # one tensor per meta spec, like the reference's *meta
            o = 0
            for spec in ld.storage._meta_specs:
                w = int(np.prod(spec.shape))
                res.append(meta[:, o:o + w].reshape((B,) + tuple(spec.shape)))
                o += w
        return tuple(res)


class ArenaIterator:
    """Iterator over an already-filled ReplayEngine (no directory behind it): what bench.py and callers that
    ingest datasets themselves use. Same next()/sample_into() contract as DeviceReplayIterator."""

    def __init__(self, engine, batch_size, nstep, discount, sampler='philox'):
        self.engine, self.batch_size, self.nstep, self.discount = engine, batch_size, nstep, discount
        self.sampler = {'mt19937': L.SAMPLER_MT19937, 'philox': L.SAMPLER_PHILOX}[sampler]

    def __iter__(self):
        return self

    def sample_into(self, out, batch=None):
        self.engine.sample_into(out, batch or self.batch_size, self.nstep, self.discount, self.sampler)

    def __next__(self):
        return self.engine.sample(self.batch_size, self.nstep, self.discount, self.sampler)


def make_replay_loader(storage, max_size, batch_size, num_workers, save_snapshot, nstep=None, discount=None, **kw):
    """replay_buffer.py:260-261 signature. Also accepts the 6-argument offline call shape that
    train_offline.py:90-93 uses — (env, replay_dir, max_size, batch_size, num_workers, discount) — which the
    reference's own 7-parameter function rejects."""
    if discount is None and isinstance(max_size, (str, os.PathLike)):
        env, replay_dir, max_size, batch_size, num_workers, discount = (storage, max_size, batch_size, num_workers,
                                                                       save_snapshot, nstep)
        return make_offline_replay_loader(env, replay_dir, max_size, batch_size, num_workers, discount, **kw)
    return DeviceReplayLoader(storage, max_size, batch_size, num_workers, save_snapshot, nstep, discount, **kw)


class _DirStorage:
    """Minimal storage stand-in for a directory of pre-collected episodes (offline datasets)."""

    def __init__(self, replay_dir, meta_specs=()):
        self._replay_dir = Path(replay_dir)
        self._meta_specs = tuple(meta_specs)


def make_offline_replay_loader(env, replay_dir, max_size, batch_size, num_workers, discount, relabel=None, **kw):
    """replay_buffer.py:246-258 with OfflineReplayBuffer's semantics (:45-100): ascending one-shot load up to the first episode
    that takes the size past max_size // workers, nstep = 1, files never touched. relabel=None follows the reference's intent —
    rewards are relabelled through `env` (relabel_episode, :31-42) whenever an env is given; pass relabel=False (or env=None) to
    train on the stored rewards. The reference's own class fails before loading anything (`_relable_reward` typo, :72 vs :84)."""
    if relabel is None:
        relabel = env is not None
    if relabel and env is None:
        raise ValueError('make_offline_replay_loader: relabel=True needs the env whose task defines the reward')
    return DeviceReplayLoader(_DirStorage(replay_dir), max_size, batch_size, num_workers, True, 1, discount, static=True, offline=True,
                              env=env, relabel=bool(relabel), **kw)
