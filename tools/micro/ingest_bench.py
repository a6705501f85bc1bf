"""Time to bring a directory of .npz episodes into the HBM arena, one decode thread vs the pool.

    python tools/micro/ingest_bench.py [episodes=1000] [steps=1000]      # walker shapes; writes the dataset to a temp dir first
"""
import sys
import tempfile
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from exorl_amd import replay_buffer as rb

E = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
T = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
O, A = 24, 6
with tempfile.TemporaryDirectory() as d:
    d = Path(d)
    rs = np.random.RandomState(0)
    t0 = time.perf_counter()
    for e in range(E):
        rows = T + 1
        rb.save_episode(dict(observation=rs.standard_normal((rows, O)).astype(np.float32), action=rs.uniform(-1, 1, (rows, A)).astype(np.float32),
                             reward=rs.uniform(0, 1, (rows, 1)).astype(np.float32), discount=np.ones((rows, 1), np.float32)),
                        d / f'episode_{e}_{T}.npz')
    print(f'wrote {E} episodes x {T} steps in {time.perf_counter() - t0:.1f} s', flush=True)
    for threads in (1, 16):
        loader = rb.make_offline_replay_loader(None, d, E * T, 1024, 1, 0.99, load_threads=threads)
        t0 = time.perf_counter()
        it = iter(loader)
        batch = next(it)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f'load_threads={threads:2d}: {E * T / 1e6:.1f} M transitions resident and first batch sampled in {dt:.2f} s '
              f'({E * T / dt / 1e6:.2f} M transitions/s)', flush=True)
        del it, loader
