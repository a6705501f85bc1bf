"""The plain-bf16 SPREAD = 2 anomaly, in a library built with EXORL_GEMM_EXPERIMENTS=1: exorl_gemm_bf16 with a k-image B operand under
mask 67108864 (refill spread over two regions) and 67108864 | 134217728 (the same + s_waitcnt vmcnt(0) behind every region-0 issue)."""
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
        for mask in (-1, 67108864, 67108864 | 134217728):
            lib.exorl_gemm_tune(mask)
            bad = 0
            for _ in range(20):
                c = torch.zeros(M, N, device='cuda')
                L.check(lib.exorl_gemm_bf16(0, 1, M, N, K, a.data_ptr(), K, b.data_ptr(), N, c.data_ptr(), N, bias.data_ptr() if with_bias else None, 0, 0, None))
                torch.cuda.synchronize()
                err = float((c.double() - ref - (bias.double() if with_bias else 0)).abs().max())
                bad += err > 1e-3
            print(f'{M}x{N}x{K} bias={with_bias} mask {mask}: wrong results in {bad} of 20 launches', flush=True)
lib.exorl_gemm_tune(-1)
