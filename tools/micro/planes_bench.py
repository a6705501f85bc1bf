"""Grouped GEMM on bf16 hi/lo planes (exorl_gemm_planes) at the agent's launch shapes: correctness against a float64 product of the
SAME planes, and time per launch by variant (exorl_gemm_tune bits: 262144 = previous kernels, 0 = 128 x TN / k32 / XCD-local,
1048576 = the same in id order, 524288 = 128 x 128 everywhere, 2097152 = 32-wide stages also for the row-image (forward) launches, whose default is 64-wide x 2).   python tools/micro/planes_bench.py [check]"""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from exorl_amd import _lib as L

lib = L.load()
C = L.C


def split(x):
    hi = x.to(torch.bfloat16)
    lo = (x - hi.float()).to(torch.bfloat16)
    return hi, lo


def make(count, a_layouts, bl, M, N, K, x3, seed=0):
    g = torch.Generator(device='cuda').manual_seed(seed)
    ps = []
    for i in range(count):
        A = torch.randn((M, K) if a_layouts[i] == 0 else (K, M), device='cuda', generator=g)
        B = torch.randn((N, K) if bl == 0 else (K, N), device='cuda', generator=g)
        ps.append((split(A), split(B), torch.zeros(M, N, device='cuda')))
    return ps


def launch(ps, a_layouts, bl, M, N, K, x3, relu=0):
    n = len(ps)
    arr = lambda xs: (C.c_void_p * n)(*[x.data_ptr() for x in xs])
    ah, al = arr([p[0][0] for p in ps]), arr([p[0][1] for p in ps])
    bh, bl_ = arr([p[1][0] for p in ps]), arr([p[1][1] for p in ps])
    cs = arr([p[2] for p in ps])
    lay = (C.c_int32 * n)(*a_layouts)
    lda = K if a_layouts[0] == 0 else M
    ldb = K if bl == 0 else N
    L.check(lib.exorl_gemm_planes(n, lay, bl, M, N, K, ah, al if x3 else None, lda, bh, bl_ if x3 else None, ldb, cs, N, relu,
                                  torch.cuda.current_stream().cuda_stream))


def reference(p, al, bl, x3):
    (ah, alo), (bh, blo), _ = p
    A = ah.double() + (alo.double() if x3 else 0)
    B = bh.double() + (blo.double() if x3 else 0)
    A = A if al == 0 else A.t()
    B = B.t() if bl == 0 else B
    full = A @ B
    if x3:      # the kernel drops lo*lo
        full = full - (alo.double() if al == 0 else alo.double().t()) @ (blo.double().t() if bl == 0 else blo.double())
    return full


def timed(ps, a_layouts, bl, M, N, K, x3, variant, iters=40):
    lib.exorl_gemm_tune(variant)
    for _ in range(5):
        launch(ps, a_layouts, bl, M, N, K, x3)
    torch.cuda.synchronize()
    L.check(lib.exorl_profile_gemm(1))
    for _ in range(iters):
        launch(ps, a_layouts, bl, M, N, K, x3)
    cap = 4096
    fl, ms, n = np.zeros(cap, np.float64), np.zeros(cap, np.float32), C.c_int32()
    L.check(lib.exorl_profile_gemm_read(fl.ctypes.data, ms.ctypes.data, cap, C.byref(n)))
    L.check(lib.exorl_profile_gemm(0))
    lib.exorl_gemm_tune(-1)
    return float(np.median(ms[:n.value])) * 1e3


H = 1024
SHAPES = [  # (tag, count, a_layouts, b_layout, M, N, K)
    ('critic+target fwd (4)', 4, [0, 0, 0, 0], 0, 1024, H, H),
    ('wgrad+dgrad (2+2)', 4, [1, 1, 0, 0], 1, 1024, H, 1024),
    ('critic fwd (2)', 2, [0, 0], 0, 1024, H, H),
    ('critic dgrad (2)', 2, [0, 0], 1, 1024, H, H),
    ('actor wgrad+dgrad (1+1)', 2, [1, 0], 1, 1024, H, 1024),
    ('actor fwd 2B (1)', 1, [0], 0, 2048, H, H),
]

if __name__ == '__main__':
    check_only = len(sys.argv) > 1 and sys.argv[1] == 'check'
    for x3 in (True, False):
        for tag, count, lay, bl, M, N, K in SHAPES:
            ps = make(count, lay, bl, M, N, K, x3)
            for variant in (0, 524288, 2097152):
                lib.exorl_gemm_tune(variant)
                for p in ps:
                    p[2].zero_()
                launch(ps, lay, bl, M, N, K, x3)
                torch.cuda.synchronize()
                worst = 0.0
                for i, p in enumerate(ps):
                    ref = reference(p, lay[i], bl, x3)
                    err = float((p[2].double() - ref).abs().max() / ref.abs().max())
                    worst = max(worst, err)
                assert worst < 2e-6, (tag, x3, variant, worst)
            lib.exorl_gemm_tune(-1)
            if check_only:
                print(f'{"x3" if x3 else "bf16":5s} {tag:26s} ok (max rel err {worst:.1e})', flush=True)
                continue
            ts = {v: timed(ps, lay, bl, M, N, K, x3, v) for v in (262144, 0, 1048576, 524288, 2097152)}
            fl = 2.0 * M * N * K * count
            print(f'{"x3" if x3 else "bf16":5s} {tag:26s} old {ts[262144]:6.2f} us | new {ts[0]:6.2f} us ({fl / ts[0] / 1e6:6.0f} TF/s) | new, id order '
                  f'{ts[1048576]:6.2f} | 128x128 everywhere {ts[524288]:6.2f} | fwd on k32 stages {ts[2097152]:6.2f}   (err {worst:.1e})', flush=True)
