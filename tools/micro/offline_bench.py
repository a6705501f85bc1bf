"""update()/s of the offline agents at the shapes BASELINE.json's other configs name (HBM replay + Philox sampler, one GPU):

    python tools/micro/offline_bench.py cql 78 12 1024            # configs[2]: CQL, quadruped_run shapes
    python tools/micro/offline_bench.py td3 17 6 512              # configs[4]: TD3 cheetah_run, the per-GPU share of batch 4096 on 8 GPUs
    python tools/micro/offline_bench.py cql 78 12 1024 bf16x3 --roofline     # + a roofline JSON line for the dominant kernel (HIP events per GEMM launch)
    python tools/micro/offline_bench.py cql 78 12 1024 bf16x3 --eager        # eager launches (for rocprofv3 --kernel-trace + tools/prof_summary.py)
"""
import json
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
precisions = (sys.argv[5] if len(sys.argv) > 5 and not sys.argv[5].startswith('--') else 'fp32,bf16x3,bf16').split(',')
ROOFLINE, EAGER = '--roofline' in sys.argv, '--eager' in sys.argv
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
    graph = False if EAGER else ag.enable_graph(it)
    n, w = (100, 20) if EAGER else (500, 50)
    for i in range(w):
        ag.update(it, i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        ag.update(it, w + i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f'{kind} O={O} A={A} B={B} {prec:7s} graph={graph}: {n / dt:8.1f} update()/s  {1e3 * dt / n:7.3f} ms', flush=True)
    if ROOFLINE:
        # the dominant kernel's launches bracketed by HIP events on the stream they run on (an instrumented eager pass of the same loop,
        # as bench.py does for the headline): algorithmic FLOPs = 2 M N K per problem, the 0.7 x event-bracket calibration of bench.py
        from exorl_amd import _lib as L
        lib = L.load()
        ag.disable_graph()
        L.check(lib.exorl_profile_gemm(1))
        nprof = 20
        for i in range(nprof):
            ag.update(it, w + n + i)
        cap = 1 << 15
        fl, ms, cnt = np.zeros(cap, np.float64), np.zeros(cap, np.float32), L.C.c_int32()
        L.check(lib.exorl_profile_gemm_read(fl.ctypes.data, ms.ctypes.data, cap, L.C.byref(cnt)))
        L.check(lib.exorl_profile_gemm(0))
        ovh = L.C.c_float()
        L.check(lib.exorl_profile_event_overhead(L.C.byref(ovh), L.current_stream()))
        fl, ms = fl[:cnt.value], np.maximum(ms[:cnt.value] - 0.7 * ovh.value, 1e-4)
        big = fl >= 2.0 * 2 * B * H * H * 0.99                      # the H x H launches (2+ problems of >= B rows)
        peak = {'bf16': 2500.0, 'bf16x3': 2500.0, 'fp32': 157.3}[prec]
        ach = float(fl[big].sum() / (ms[big].sum() * 1e-3) / 1e12)
        print(json.dumps({'workload': f'{kind} O={O} A={A} B={B} H={H}', 'dtype': prec, 'ms_per_step': 1e3 * dt / n, 'updates_per_s': n / dt,
                          'gemm_gflop_per_step': float(fl.sum() / nprof / 1e9), 'gemm_us_per_step': float(ms.sum() * 1e3 / nprof),
                          'gemm_launches_per_step': cnt.value / nprof,
                          'roofline': {'bound': 'mfma', 'achieved': ach, 'peak': peak, 'unit': 'TFLOP/s', 'frac': ach / peak, 'traffic': None,
                                       'kernel': 'gemm16p_kernel / gemm16p_mixed_kernel on the H x H launches', 'launches': int(big.sum()),
                                       'avg_us': float(ms[big].mean() * 1e3), 'flop_per_launch': float(fl[big].mean())}}), flush=True)
    del ag, it
