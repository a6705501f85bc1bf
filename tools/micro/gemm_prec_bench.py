"""Times exorl_gemm (fp32 operands in memory) in its three MFMA precisions on the agent's layer shapes."""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from exorl_amd import _lib as L

lib = L.load()
H = 1024


def run(prec, al, bl, M, N, K, iters=30):
    a = torch.randn(M * K, device='cuda')
    b = torch.randn(K * N, device='cuda')
    c = torch.empty(M, N, device='cuda')
    lda = K if al == 0 else M
    ldb = K if bl == 0 else N
    st = torch.cuda.current_stream().cuda_stream

    def launch():
        L.check(lib.exorl_gemm(prec, al, bl, M, N, K, a.data_ptr(), lda, b.data_ptr(), ldb, c.data_ptr(), N, None, 0, 0, st))
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
    return float(np.median(ms[:n.value])) * 1e3


for (al, bl, M, N, K, tag) in [(0, 0, 2048, H, H, 'fwd 2048'), (0, 0, 1024, H, H, 'fwd 1024'), (0, 1, 2048, H, H, 'dgrad'), (1, 1, H, H, 2048, 'wgrad')]:
    for prec, nm in ((0, 'fp32'), (1, 'bf16'), (2, 'bf16x3')):
        us = run(prec, al, bl, M, N, K)
        print(f'{tag:10s} {nm:8s} {us:8.2f} us  {2.0 * M * N * K / us / 1e6:8.1f} TFLOP/s', flush=True)
