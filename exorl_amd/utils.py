"""Host-side helpers with the reference's names and meaning (/root/reference/utils/utils.py), for the
callers of the hot path (pretrain.py / train_offline.py use utils.eval_mode, utils.schedule, utils.set_seed_everywhere ...).
The tensor math these names used to do on the update path now lives in libexorl_hip.so."""
import random
import re

import numpy as np
import torch


class eval_mode:
    """utils.py:15-28 — temporarily puts objects with .training/.train() in eval mode."""

    def __init__(self, *models):
        self.models = models

    def __enter__(self):
        self.prev_states = [m.training for m in self.models]
        for m in self.models:
            m.train(False)

    def __exit__(self, *args):
        for m, state in zip(self.models, self.prev_states):
            m.train(state)
        return False


def set_seed_everywhere(seed):
    """utils.py:31-36."""
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    np.random.seed(seed)
    random.seed(seed)


def schedule(schdl, step):
    """utils.py:199-219: constant | linear(init,final,duration) | step_linear(init,f1,d1,f2,d2)."""
    try:
        return float(schdl)
    except ValueError:
        m = re.match(r'linear\((.+),(.+),(.+)\)', schdl)
        if m:
            init, final, duration = (float(g) for g in m.groups())
            mix = float(np.clip(step / duration, 0.0, 1.0))
            return (1.0 - mix) * init + mix * final
        m = re.match(r'step_linear\((.+),(.+),(.+),(.+),(.+)\)', schdl)
        if m:
            init, final1, duration1, final2, duration2 = (float(g) for g in m.groups())
            if step <= duration1:
                mix = float(np.clip(step / duration1, 0.0, 1.0))
                return (1.0 - mix) * init + mix * final1
            mix = float(np.clip((step - duration1) / duration2, 0.0, 1.0))
            return (1.0 - mix) * final1 + mix * final2
    raise NotImplementedError(schdl)


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
