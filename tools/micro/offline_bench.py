"""update()/s of the offline agents at the shapes BASELINE.json's other configs name (HBM replay + Philox sampler, one GPU):

    python tools/micro/offline_bench.py cql 78 12 1024            # configs[2]: CQL, quadruped_run shapes
    python tools/micro/offline_bench.py td3 17 6 512              # configs[4]: TD3 cheetah_run, the per-GPU share of batch 4096 on 8 GPUs
"""
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from exorl_amd import agents
from exorl_amd.engine import ReplayEngine
from exorl_amd.replay_buffer import ArenaIterator

kind, O, A, B = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
precisions = (sys.argv[5] if len(sys.argv) > 5 else 'fp32,bf16x3,bf16').split(',')
H, EPISODES, EP_LEN = 1024, 300, 1000


def make(precision):
    if kind == 'cql':
        return agents.CQLAgent('cql', (O,), (A,), 'cuda', 1e-4, H, 0.01, 1, B, False, 0.01, 3, 5.0, False, precision=precision)
    if kind == 'td3':
        return agents.TD3Agent('td3', (O,), (A,), 'cuda', 1e-4, H, 0.01, 0.2, 1, B, 0.3, False, precision=precision)
    if kind == 'crr':
        return agents.CRRAgent('crr', (O,), (A,), 'cuda', 1e-4, H, 0.01, 10, 'indicator', 0.2, 1, B, 0.3, False, precision=precision)
    if kind == 'bc':
        return agents.BCAgent('bc', (O,), (A,), 'cuda', 1e-4, H, B, 0.2, False, precision=precision)
    return agents.TD3BCAgent('td3_bc', (O,), (A,), 'cuda', 1e-4, H, 0.01, 0.2, 1, B, 0.3, False, 2.5, precision=precision)


eng = ReplayEngine((O,), np.float32, A, 0, EPISODES * (EP_LEN + 1) + 64, EPISODES + 8, 'cuda')
slots = []
for e in range(EPISODES):
    rs = np.random.RandomState(11 + e)
    rows = EP_LEN + 1
    slots.append(eng.append_episode(dict(observation=rs.standard_normal((rows, O)).astype(np.float32),
                                         action=rs.uniform(-1, 1, (rows, A)).astype(np.float32),
                                         reward=rs.uniform(0, 1, (rows, 1)).astype(np.float32), discount=np.ones((rows, 1), np.float32))))
eng.set_order(slots)
for prec in precisions:
    torch.manual_seed(1)
    eng.seed_philox(2)
    ag = make(prec)
    it = ArenaIterator(eng, B, 1, 0.99, 'philox')
    graph = ag.enable_graph(it)
    n, w = 500, 50
    for i in range(w):
        ag.update(it, i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        ag.update(it, w + i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f'{kind} O={O} A={A} B={B} {prec:7s} graph={graph}: {n / dt:8.1f} update()/s  {1e3 * dt / n:7.3f} ms', flush=True)
    del ag, it
