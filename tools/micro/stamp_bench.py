"""In-kernel clock and phase split of the forward H x H GEMM (split-bf16, 128 x 128 and 128 x 64 tiles): the stamped diagnostic build
(exorl_gemm_tune bit 33554432) is launched back to back for ~2 s on random data, then the last launch's stamps are read.
clock = d(s_memtime) / d(s_memrealtime) x 100 MHz (MI355X_MICROARCH.md, DVFS give-back (6)).     python tools/micro/stamp_bench.py"""
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
sys.path.insert(0, str(Path(__file__).resolve().parent))
from planes_bench import make, launch, lib, H
from exorl_amd import _lib as L

STAMP = 33554432
K32 = 2097152          # forward launches on k32 x 4 stages instead of k64 x 2
A = 1 << 26
SP = 536870912
CASES = [('fwd x4 128x128 k64x2', 4, [0, 0, 0, 0], 0, 0), ('  .. refill in one region (old)', 4, [0, 0, 0, 0], 0, SP), ('  .. old, no DMA after prologue', 4, [0, 0, 0, 0], 0, A), ('  .. old, no fragment reads', 4, [0, 0, 0, 0], 0, 2 * A),
         ('  .. old, no MFMA', 4, [0, 0, 0, 0], 0, 3 * A), ('fwd x4 128x128 k32x4', 4, [0, 0, 0, 0], 0, K32), ('fwd x2 128x64 k64x2', 2, [0, 0], 0, 0),
         ('fwd x2 128x64 k32x4', 2, [0, 0], 0, K32), ('dgrad x4 128x128 k32x4', 4, [0, 0, 0, 0], 1, 0), ('dgrad x2 128x64 k32x4', 2, [0, 0], 1, 0),
         ('wgrad x4 128x128 k32x4', 4, [1, 1, 1, 1], 1, 0)]
import os
if os.environ.get('EXORL_GEMM_EXPERIMENTS'):      # library built with the experiments: the 8-wave forward kernels
    CASES += [('fwd x4 128x128 k64x2, 8 waves', 4, [0, 0, 0, 0], 0, 268435456), ('fwd x2 128x64 k64x2, 8 waves', 2, [0, 0], 0, 268435456)]
M = N = K = H
for tag, count, lay, bl, extra in CASES:
    ps = make(count, lay, bl, M, N, K, True)
    lib.exorl_gemm_tune(STAMP | extra)
    t0 = time.time()
    while time.time() - t0 < 1.5:
        for _ in range(200):
            launch(ps, lay, bl, M, N, K, True)
        torch.cuda.synchronize()
    buf = np.zeros(32 * 256, np.uint64)
    L.check(lib.exorl_debug_gemm_stamps(buf.ctypes.data, buf.size))
    lib.exorl_gemm_tune(-1)
    st = buf.reshape(256, 4, 8).astype(np.int64)          # [workgroup][wave][word]
    t = st[:, :, :4]
    loop_us = (st[:, :, 5] - st[:, :, 4]) / 100.0
    loop_cyc = (t[:, :, 2] - t[:, :, 1]).astype(np.float64)
    clk = np.median(loop_cyc / loop_us / 1e3)
    med = lambda x: float(np.median(x))
    nmf = 1024 // 16 * (6 if '128x64' in tag else 12) * 32
    print(f'{tag:32s} k-loop {med(loop_cyc):7.0f} cyc = {med(loop_us):5.2f} us at {clk:.2f} GHz (MFMA {nmf}) | waits per wave: own DMA {med(st[:, :, 6]):6.0f} cyc, '
          f'barrier {med(st[:, :, 7]):6.0f} cyc (max wave {st[:, :, 7].max(axis=1).mean():6.0f}) | prologue {med(t[:, :, 1] - t[:, :, 0]):5.0f} cyc | '
          f'epilogue {med(t[:, :, 3] - t[:, :, 2]):5.0f} cyc', flush=True)
