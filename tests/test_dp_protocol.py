"""Data-parallel protocol on CPU: two gloo ranks, each holding half of a global batch, sum-all-reduce
the three batch-global quantities (critic grads, sum|Q| for lambda, actor grads) and must land on the same
parameters and metrics as the single-process large-batch update. The compute engine here is the oracle — the
HIP engine runs the same phase split (tests/test_gpu_agent.py::test_virtual_ranks_equal_single_rank)."""
import os
import sys
import tempfile
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / 'tests'))


def _worker(rank, world, kind, init_file, out_dir):
    import _synth
    from oracle.agents import OracleAgent, param_shapes
    dist.init_process_group('gloo', init_method=f'file://{init_file}', rank=rank, world_size=world)

    def allreduce(arrays):
        for a in arrays:
            t = torch.from_numpy(a)          # shares memory: summed in place
            dist.all_reduce(t)
    O, A, H, B = 7, 3, 32, 16
    ash, csh = param_shapes(kind, O, A, H)
    pa = list(_synth.synth_params(ash, 1).values())
    pc = list(_synth.synth_params(csh, 2).values()) if csh else None
    ag = OracleAgent(kind, pa, pc, world_size=world, allreduce=allreduce)
    ns = _synth.NoiseStream(5)
    metrics = []
    for i in range(3):
        step = 2 * i if kind == 'ddpg' else i
        full = _synth.synth_batch(3, i, B, O, A)
        n1, n2 = ns.draw((B, A)), ns.draw((B, A))
        sl = slice(rank * B // world, (rank + 1) * B // world)
        shard = tuple(x[sl] for x in full)
        metrics.append(ag.update(shard, step, n1[sl], n2[sl]))
    np.savez(Path(out_dir) / f'rank{rank}.npz', *ag.actor, *(ag.critic if pc else []))
    if rank == 0:
        np.save(Path(out_dir) / 'metrics.npy', np.array([[m[k] for k in sorted(m)] for m in metrics]))
    dist.destroy_process_group()


@pytest.mark.parametrize('kind', ['td3_bc', 'ddpg', 'bc'])
def test_two_ranks_equal_single_process(kind):
    import _synth
    from oracle.agents import OracleAgent, param_shapes
    world = 2
    with tempfile.TemporaryDirectory() as td:
        init_file = os.path.join(td, 'rendezvous')
        mp.spawn(_worker, args=(world, kind, init_file, td), nprocs=world, join=True)
        r0, r1 = np.load(Path(td) / 'rank0.npz'), np.load(Path(td) / 'rank1.npz')
        dp_metrics = np.load(Path(td) / 'metrics.npy')
        O, A, H, B = 7, 3, 32, 16
        ash, csh = param_shapes(kind, O, A, H)
        ag = OracleAgent(kind, list(_synth.synth_params(ash, 1).values()),
                         list(_synth.synth_params(csh, 2).values()) if csh else None)
        ns = _synth.NoiseStream(5)
        ref_metrics = []
        for i in range(3):
            step = 2 * i if kind == 'ddpg' else i
            m = ag.update(_synth.synth_batch(3, i, B, O, A), step, ns.draw((B, A)), ns.draw((B, A)))
            ref_metrics.append([m[k] for k in sorted(m)])
        ref = ag.actor + (ag.critic if csh else [])
        for i, p in enumerate(ref):
            a0, a1 = r0[f'arr_{i}'], r1[f'arr_{i}']
            assert np.array_equal(a0, a1), f'ranks diverged on tensor {i}'          # replicas stay bit-identical
            np.testing.assert_allclose(a0, p, rtol=2e-5, atol=2e-7, err_msg=f'tensor {i}')   # == large-batch update
        np.testing.assert_allclose(dp_metrics, np.array(ref_metrics), rtol=2e-5, atol=1e-6)
