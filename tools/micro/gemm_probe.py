"""Bottleneck probes for the LDS-DMA bf16 GEMM: K sweep (per-k-tile cost vs fixed cost), cache-resident operands
(lda = ldb = 0: every row aliases one 2 KB line set), grid-size sweep."""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from exorl_amd import _lib as L

lib = L.load()
st = torch.cuda.current_stream().cuda_stream


def timed(al, bl, M, N, K, lda=None, ldb=None, iters=40):
    a = torch.randn(max(M, K) * max(M, K), device='cuda').to(torch.bfloat16)
    b = torch.randn(max(N, K) * max(N, K), device='cuda').to(torch.bfloat16)
    c = torch.empty(M, N, device='cuda')
    lda = (K if al == 0 else M) if lda is None else lda
    ldb = (K if bl == 0 else N) if ldb is None else ldb

    def launch():
        L.check(lib.exorl_gemm_bf16(al, bl, M, N, K, a.data_ptr(), lda, b.data_ptr(), ldb, c.data_ptr(), N, None, 0, 0, st))
    for _ in range(5):
        launch()
    torch.cuda.synchronize()
    L.check(lib.exorl_profile_gemm(1))
    for _ in range(iters):
        launch()
    cap = 4096
    fl, ms, n = np.zeros(cap, np.float64), np.zeros(cap, np.float32), L.C.c_int32()
    L.check(lib.exorl_profile_gemm_read(fl.ctypes.data, ms.ctypes.data, cap, L.C.byref(n)))
    L.check(lib.exorl_profile_gemm(0))
    return float(np.median(ms[:n.value])) * 1e3, float(np.min(ms[:n.value])) * 1e3


lib.exorl_gemm_tune(0)
print('K sweep (M=N=1024, fwd):')
for K in (64, 128, 256, 512, 1024, 2048, 4096):
    print(f'  K={K:5d}  median {timed(0, 0, 1024, 1024, K)[0]:7.2f} us', flush=True)
print('aliased operands (lda=ldb=0 -> L1/L2 resident), M=N=1024:')
for K in (256, 1024, 4096):
    print(f'  K={K:5d}  median {timed(0, 0, 1024, 1024, K, 0, 0)[0]:7.2f} us', flush=True)
print('grid sweep K=1024 fwd:')
for M, N in ((64, 64), (256, 256), (512, 512), (1024, 512), (1024, 1024), (2048, 1024), (4096, 1024), (4096, 4096)):
    us = timed(0, 0, M, N, 1024)[0]
    print(f'  M={M:5d} N={N:5d} tiles={M * N // 4096:5d}  {us:7.2f} us  {2.0 * M * N * 1024 / us / 1e6:7.1f} TF', flush=True)
