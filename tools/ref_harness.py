"""Loads the reference's hot-path files (read-only, /root/reference) for fixture generation.

Only used by tools/gen_golden.py in the build container; never imported by the product,
tests, bench or smoke (the reference does not exist on the GPU box).
Recipe: SURVEY.md Appendix B (stub modules for imports that are unused on this path).
"""
import importlib
import importlib.util
import sys
import types

import numpy as np

REF = '/root/reference'


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _Array:  # stand-in for dm_env.specs.Array (shape, dtype, name)
    def __init__(self, shape, dtype, name=None):
        self.shape, self.dtype, self.name = tuple(shape), np.dtype(dtype), name


def _load(name, path):
    sp = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(sp)
    sys.modules[name] = m
    sp.loader.exec_module(m)
    return m


def load_reference():
    _stub('hydra')
    _stub('omegaconf', OmegaConf=object)
    r = _stub('dm_control.utils.rewards')
    u = _stub('dm_control.utils', rewards=r)
    _stub('dm_control', utils=u)
    _stub('dm_env', specs=_stub('dm_env.specs', Array=_Array))
    ref = types.SimpleNamespace()
    ref.Array = _Array
    ref.rb = _load('ref_replay_buffer', f'{REF}/utils/replay_buffer.py')
    ref.utils = _load('utils', f'{REF}/utils/utils.py')
    for n in ('td3_bc', 'td3', 'bc', 'cql', 'crr'):
        setattr(ref, n, _load(f'ref_{n}', f'{REF}/agents/offline_learning/{n}.py'))
    pkg = types.ModuleType('refunsup')
    pkg.__path__ = [f'{REF}/agents/unsupervised_learning']
    sys.modules['refunsup'] = pkg
    for n in ('ddpg', 'proto', 'rnd', 'icm', 'icm_apt', 'disagreement', 'diayn', 'aps', 'smm'):
        setattr(ref, n, importlib.import_module(f'refunsup.{n}'))
    return ref


if __name__ == '__main__':
    ref = load_reference()
    import torch
    torch.manual_seed(0)
    ag = ref.td3_bc.TD3BCAgent('td3_bc', (24,), (6,), 'cpu', 1e-4, 64, 0.01, 0.2, 1, 8, 0.3, True, 2.5)
    b = (np.random.randn(8, 24).astype(np.float32), np.random.uniform(-1, 1, (8, 6)).astype(np.float32),
         np.random.rand(8, 1).astype(np.float32), np.ones((8, 1), np.float32), np.random.randn(8, 24).astype(np.float32))
    print(ag.update(iter([b]), 0))
