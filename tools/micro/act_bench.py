"""Latency of act() on one observation: device time of the single-launch kernel (HIP events around 200 back-to-back calls) and
end-to-end microseconds per call from Python (numpy observation in -> numpy action out, including the stream synchronise), against the generic
multi-launch path (exorl_gemm_tune bit 256 routes exorl_agent_act back to net_forward + head + sampling + copy)."""
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from exorl_amd import _lib as L, agents  # noqa: E402

lib = L.load()
O, A, H, B = 24, 6, 1024, 1024
for precision in ('fp32', 'bf16x3'):
    ag = agents.TD3BCAgent('td3_bc', (O,), (A,), 'cuda', 1e-4, H, 0.01, '0.2', 1, B, 0.3, False, 2.5, precision=precision)
    ag.num_expl_steps = 0
    obs = np.random.RandomState(0).standard_normal(O).astype(np.float32)
    for eval_mode in (True, False):
        for label, tune in (('one launch (exorl_agent_act_host)', -1), ('generic path (net_forward + head + sample + copies)', 256)):
            lib.exorl_gemm_tune(tune)
            fast = tune < 0
            call = (lambda: ag.act(obs, 10, eval_mode)) if fast else (lambda: ag.engine.act(obs, 0.2, eval_mode).cpu().numpy()[0])
            for _ in range(50):
                call()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 2000
            for _ in range(n):
                call()
            e2e = (time.perf_counter() - t0) / n * 1e6
            # device time: events around back-to-back launches through the raw entry (no synchronise in between)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            od = torch.from_numpy(obs).cuda()
            out = torch.empty(A, device='cuda')
            st = L.current_stream()
            m = 200
            e0.record()
            for _ in range(m):
                L.check(lib.exorl_agent_act(ag.engine.h, od.data_ptr(), 1, 0.2, int(eval_mode), None, out.data_ptr(), st))
            e1.record()
            torch.cuda.synchronize()
            dev = e0.elapsed_time(e1) / m * 1e3
            print(f'{precision:7s} eval_mode={eval_mode!s:5s} {label:55s}: {e2e:7.1f} us per act() from Python, {dev:6.1f} us per call on the device stream', flush=True)
    lib.exorl_gemm_tune(-1)
    del ag

# ---- pixel observations (config 4 shapes: (3, 84, 84) uint8, A = 9, feature 50, hidden 1024): encoder on one frame, then the fused trunk + policy
for precision in ('fp32', 'bf16x6'):
    pa = agents.DDPGAgent('ddpg', True, 'pixels', (3, 84, 84), (9,), 'cuda', 1e-4, 50, 1024, 0.01, 0, 2, 0.2, 3, 16, 0.3, True, False, False, precision=precision)
    frame = np.random.RandomState(1).randint(0, 256, (3, 84, 84)).astype(np.uint8)
    for label, tune in (('fused: encoder + trunk_one + policy kernel', -1), ('generic path (split-K GEMM + reduce + LN + 3 GEMMs + head + copies)', 256)):
        lib.exorl_gemm_tune(tune)
        for _ in range(20):
            pa.act(frame, {}, 10**6, True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 500
        for _ in range(n):
            pa.act(frame, {}, 10**6, True)
        e2e = (time.perf_counter() - t0) / n * 1e6
        print(f'pixels {precision:7s} eval_mode=True  {label:70s}: {e2e:7.1f} us per act() from Python', flush=True)
    lib.exorl_gemm_tune(-1)
    del pa
