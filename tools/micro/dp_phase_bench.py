"""Single-GPU cost of the data-parallel step structure: sampler + 4 update phases launched eagerly (what each rank runs between
its RCCL all-reduces) against the fused eager step and the captured graph."""
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import bench
from exorl_amd import agents, _lib as L
from exorl_amd.replay_buffer import ArenaIterator

prec = sys.argv[1] if len(sys.argv) > 1 else 'bf16x3'
replay = bench.synth_replay(0, 1, 'cuda:0')
O, A, H, B = bench.O, bench.A, bench.H, bench.B


def timeit(fn, n=1000, w=100):
    for i in range(w):
        fn(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        fn(w + i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


ag = agents.TD3BCAgent('td3_bc', (O,), (A,), 'cuda:0', 1e-4, H, 0.01, '0.2', 1, B, 0.3, False, 2.5, precision=prec, seed=1)
it = ArenaIterator(replay, B, 1, 0.99, 'philox')
print(prec, 'eager fused step  ms', timeit(lambda i: ag.update(it, i)), flush=True)


def phases(i):
    ag._load_batch(it)
    for ph in range(4):
        ag.engine.update_phase(ph, 0.2)


print(prec, 'eager 4 phases    ms', timeit(phases), flush=True)
scratch = torch.zeros(2_200_000, device='cuda')


def phases_sync(i):           # a dependent small torch kernel between phases, standing in for the collectives' stream hand-offs
    ag._load_batch(it)
    for ph in range(4):
        ag.engine.update_phase(ph, 0.2)
        if ph < 3:
            scratch[:4].add_(1.0)


print(prec, 'phases + 3 torch ops ms', timeit(phases_sync), flush=True)
assert ag.enable_graph(it)
print(prec, 'captured graph    ms', timeit(lambda i: ag.update(it, i)), flush=True)
