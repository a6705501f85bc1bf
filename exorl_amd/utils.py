"""Host-side helpers with the reference's names and meaning (/root/reference/utils/utils.py), for the
callers of the hot path (pretrain.py / train_offline.py use utils.eval_mode, utils.schedule, utils.set_seed_everywhere ...).
The tensor math these names used to do on the update path now lives in libexorl_hip.so."""
import contextlib
import functools
import random
import re
import time

import numpy as np
import torch


@contextlib.contextmanager
def eval_mode(*models):
    """`with utils.eval_mode(agent): ...` (reference utils.py:15-28): everything with a .training flag and a .train(bool) method is
    switched to eval for the duration of the block and put back the way it was, also when the block raises."""
    before = [(m, m.training) for m in models]
    try:
        for m, _ in before:
            m.train(False)
        yield
    finally:
        for m, was in before:
            m.train(was)


def set_seed_everywhere(seed):
    """Seeds the four generators the reference seeds (utils.py:31-36); the HBM replay's MT19937 sampler continues from the `random` /
    `np.random` state this leaves behind (replay_buffer.py:169,222)."""
    for seeder in (torch.manual_seed, np.random.seed, random.seed):
        seeder(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


_SCHEDULE = re.compile(r'^(linear|step_linear)\((.+)\)$')


@functools.lru_cache(maxsize=256)
def _schedule_segments(spec):
    """'linear(a,b,T)' -> [(0, T, a, b)]; 'step_linear(a,b,T1,c,T2)' -> [(0, T1, a, b), (T1, T2, b, c)]; a bare number -> a constant.
    Each segment is (start, duration, value at start, value at end)."""
    try:
        return float(spec)
    except ValueError:
        pass
    m = _SCHEDULE.match(spec)
    if m:
        v = [float(x) for x in m.group(2).split(',')]
        if m.group(1) == 'linear' and len(v) == 3:
            return ((0.0, v[2], v[0], v[1]),)
        if m.group(1) == 'step_linear' and len(v) == 5:
            return ((0.0, v[2], v[0], v[1]), (v[2], v[4], v[1], v[3]))
    raise NotImplementedError(spec)


def schedule(schdl, step):
    """The stddev schedule grammar of the agent YAMLs (utils.py:199-219) as a table of linear segments: the value at `step` on the segment
    that contains it (a step ON a boundary belongs to the segment that ends there), clamped past the last one. Per segment the same
    double-precision expression as the reference, (1 - mix) * v0 + mix * v1 with mix = clip((step - start) / duration, 0, 1), so the
    values are bit-identical (tests/golden/utils_g2.npz)."""
    segs = _schedule_segments(schdl if isinstance(schdl, str) else repr(float(schdl)))
    if isinstance(segs, float):
        return segs
    start, duration, v0, v1 = next((sg for sg in segs if step <= sg[0] + sg[1]), segs[-1])
    mix = min(max((step - start) / duration, 0.0), 1.0)
    return (1.0 - mix) * v0 + mix * v1


def to_torch(xs, device):
    """utils.py:55-56."""
    return tuple(torch.as_tensor(x, device=device) for x in xs)


def hard_update_params(net, target_net):
    """utils.py:50-52 (init_from, snapshot interchange)."""
    for p, t in zip(net.parameters(), target_net.parameters()):
        t.data.copy_(p.data)
    if getattr(target_net, '_on_change', None):
        target_net._on_change()


def soft_update_params(net, target_net, tau):
    """utils.py:44-47 on NetView parameter views -> exorl_soft_update per tensor."""
    from . import _lib as L
    lib = L.load()
    for p, t in zip(net.parameters(), target_net.parameters()):
        L.check(lib.exorl_soft_update(p.data_ptr(), t.data_ptr(), p.numel(), tau, L.current_stream()))
    if getattr(target_net, '_on_change', None):
        target_net._on_change()


class _StepGate:
    """Shared arithmetic of the training loops' step predicates (pretrain.py:209-215, train_offline.py:84-88): the configured frame count is
    converted to agent steps by integer division with `action_repeat`, at call time; a count of None switches the gate off."""

    def __init__(self, frames, action_repeat=1):
        self.frames, self.action_repeat = frames, action_repeat

    def _steps(self):
        return None if self.frames is None else self.frames // self.action_repeat


class Until(_StepGate):
    """utils.py:87-96: `while train_until_step(step)` — true while step < until // action_repeat; always true when until is None."""

    def __call__(self, step):
        limit = self._steps()
        return True if limit is None else step < limit


class Every(_StepGate):
    """utils.py:99-110: `if eval_every_step(step)` — true on multiples of every // action_repeat; never when every is None."""

    def __call__(self, step):
        period = self._steps()
        return False if period is None else step % period == 0


class Timer:
    """utils.py:113-125: reset() -> (seconds since the previous reset, seconds since construction); total_time()."""

    def __init__(self):
        self._t0 = self._lap = time.time()

    def reset(self):
        now = time.time()
        lap, self._lap = now - self._lap, now
        return lap, now - self._t0

    def total_time(self):
        return time.time() - self._t0
