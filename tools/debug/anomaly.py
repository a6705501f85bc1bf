"""Round 2's plain-bf16 SPREAD = 2 anomaly as a regression script for the DEFAULT build (round 3: cause found and fixed): exorl_gemm_bf16
with a k-image B operand, 20 launches per shape under the default schedule (refill spread over two regions) and under bit 536870912 (refill in one
region), counted against the float64 product. Before the fix the default line read "wrong results in 20 of 20 launches" at 1024^3."""
import sys
from pathlib import Path
import numpy as np
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from exorl_amd import _lib as L
lib = L.load()
rs = np.random.RandomState(0)
for M, N, K in ((128, 128, 256), (256, 384, 512), (1024, 1024, 1024)):
    a = torch.from_numpy(rs.standard_normal((M, K)).astype(np.float32)).cuda().to(torch.bfloat16).contiguous()
    b = torch.from_numpy(rs.standard_normal((K, N)).astype(np.float32)).cuda().to(torch.bfloat16).contiguous()
    bias = torch.from_numpy(rs.standard_normal(N).astype(np.float32)).cuda()
    ref = a.double() @ b.double()
    for with_bias in (False, True):
        for mask in (-1, 536870912):
            lib.exorl_gemm_tune(mask)
            bad, ms = 0, []
            for _ in range(20):
                c = torch.zeros(M, N, device='cuda')
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                L.check(lib.exorl_gemm_bf16(0, 1, M, N, K, a.data_ptr(), K, b.data_ptr(), N, c.data_ptr(), N, bias.data_ptr() if with_bias else None, 0, 0, None))
                e1.record()
                torch.cuda.synchronize()
                ms.append(e0.elapsed_time(e1))
                err = float((c.double() - ref - (bias.double() if with_bias else 0)).abs().max())
                bad += err > 1e-3
            print(f'{M}x{N}x{K} bias={with_bias} schedule {"two-region refill (default)" if mask < 0 else "one-region refill"}: wrong results in {bad} of 20 launches, '
                  f'median {sorted(ms)[10] * 1e3:.1f} us', flush=True)
lib.exorl_gemm_tune(-1)
