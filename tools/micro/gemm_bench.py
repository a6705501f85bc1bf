"""Times the bf16-operand GEMM variants (exorl_gemm_tune bits) on the agent's layer shapes with HIP events."""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from exorl_amd import _lib as L

lib = L.load()
H = 1024


def run(al, bl, M, N, K, variant, iters=50):
    lib.exorl_gemm_tune(variant)
    a = torch.randn(M * K, device='cuda').to(torch.bfloat16)
    b = torch.randn(K * N, device='cuda').to(torch.bfloat16)
    c = torch.empty(M, N, device='cuda')
    lda = K if al == 0 else M
    ldb = K if bl == 0 else N
    st = torch.cuda.current_stream().cuda_stream

    def launch():
        L.check(lib.exorl_gemm_bf16(al, bl, M, N, K, a.data_ptr(), lda, b.data_ptr(), ldb, c.data_ptr(), N, None, 0, 0, st))
    for _ in range(5):
        launch()
    torch.cuda.synchronize()
    L.check(lib.exorl_profile_gemm(1))          # per-launch HIP events on the launch stream (device time, not host pacing)
    for _ in range(iters):
        launch()
    cap = 4096
    fl, ms, n = np.zeros(cap, np.float64), np.zeros(cap, np.float32), L.C.c_int32()
    L.check(lib.exorl_profile_gemm_read(fl.ctypes.data, ms.ctypes.data, cap, L.C.byref(n)))
    L.check(lib.exorl_profile_gemm(0))
    return float(np.median(ms[:n.value])) * 1e3


names = {0: 'glds 64x64 s4', 4096: 'glds 64x64 s3', 8192: 'glds 64x64 s2'}
for (al, bl, M, N, K, tag) in [(0, 0, 2048, H, H, 'fwd 2048'), (0, 0, 1024, H, H, 'fwd 1024'), (0, 0, 4096, H, H, 'fwd 4096'), (0, 1, 1024, H, H, 'dgrad'),
                               (1, 1, H, H, 1024, 'wgrad')]:
    for v, nm in names.items():
        us = run(al, bl, M, N, K, v)
        print(f'{tag:10s} {nm:16s} {us:8.2f} us  {2.0 * M * N * K / us / 1e6:8.1f} TFLOP/s', flush=True)
