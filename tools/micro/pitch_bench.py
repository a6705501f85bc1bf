"""Does the row pitch of the operand planes matter to the H x H GEMM?  The agent's planes are [rows][1024] bf16: every row starts 2 KB after
the previous one, so the 8-16 row segments one LDS-DMA instruction fetches sit at a power-of-two stride. Same launches with the planes
allocated at a padded pitch (lda = K + pad): time per launch by pad.   python tools/micro/pitch_bench.py"""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
sys.path.insert(0, str(Path(__file__).resolve().parent))
from planes_bench import split, reference, lib, H
from exorl_amd import _lib as L
C = L.C


def make(count, a_layouts, bl, M, N, K, pad, seed=0):
    g = torch.Generator(device='cuda').manual_seed(seed)
    ps = []
    for i in range(count):
        ar, ac = (M, K) if a_layouts[i] == 0 else (K, M)
        br, bc = (N, K) if bl == 0 else (K, N)
        A = torch.randn(ar, ac, device='cuda', generator=g)
        B = torch.randn(br, bc, device='cuda', generator=g)
        def planes(X):
            hi, lo = split(X)
            out = []
            for p in (hi, lo):
                buf = torch.zeros(X.shape[0], X.shape[1] + pad, dtype=torch.bfloat16, device='cuda')
                buf[:, :X.shape[1]] = p
                out.append(buf)
            return out
        ps.append((planes(A), planes(B), torch.zeros(M, N, device='cuda'), (split(A), split(B))))
    return ps


def launch(ps, a_layouts, bl, M, N, K, pad):
    n = len(ps)
    arr = lambda xs: (C.c_void_p * n)(*[x.data_ptr() for x in xs])
    ah, al = arr([p[0][0] for p in ps]), arr([p[0][1] for p in ps])
    bh, bl_ = arr([p[1][0] for p in ps]), arr([p[1][1] for p in ps])
    cs = arr([p[2] for p in ps])
    lay = (C.c_int32 * n)(*a_layouts)
    lda = (K if a_layouts[0] == 0 else M) + pad
    ldb = (K if bl == 0 else N) + pad
    L.check(lib.exorl_gemm_planes(n, lay, bl, M, N, K, ah, al, lda, bh, bl_, ldb, cs, N, 0, torch.cuda.current_stream().cuda_stream))


def timed(ps, lay, bl, M, N, K, pad, iters=200):
    for _ in range(5):
        launch(ps, lay, bl, M, N, K, pad)
    torch.cuda.synchronize()
    L.check(lib.exorl_profile_gemm(1))
    for _ in range(iters):
        launch(ps, lay, bl, M, N, K, pad)
    cap = 4096
    fl, ms, n = np.zeros(cap, np.float64), np.zeros(cap, np.float32), C.c_int32()
    L.check(lib.exorl_profile_gemm_read(fl.ctypes.data, ms.ctypes.data, cap, C.byref(n)))
    L.check(lib.exorl_profile_gemm(0))
    return float(np.median(ms[:n.value])) * 1e3


SHAPES = [
    ('critic+target fwd (4)', 4, [0, 0, 0, 0], 0, 1024, H, H),
    ('wgrad+dgrad (2+2)', 4, [1, 1, 0, 0], 1, 1024, H, 1024),
    ('critic fwd (2)', 2, [0, 0], 0, 1024, H, H),
    ('critic dgrad (2)', 2, [0, 0], 1, 1024, H, H),
    ('actor wgrad+dgrad (1+1)', 2, [1, 0], 1, 1024, H, 1024),
]
for tag, count, lay, bl, M, N, K in SHAPES:
    res = {}
    for pad in (0, 8, 64, 128):
        ps = make(count, lay, bl, M, N, K, pad)
        launch(ps, lay, bl, M, N, K, pad)
        torch.cuda.synchronize()
        worst = 0.0
        for i, p in enumerate(ps):
            ref = reference((p[3][0], p[3][1], None), lay[i], bl, True)
            worst = max(worst, float((p[2].double() - ref).abs().max() / ref.abs().max()))
        assert worst < 2e-6, (tag, pad, worst)
        res[pad] = min(timed(ps, lay, bl, M, N, K, pad) for _ in range(2))
        del ps
    print(f'{tag:26s} ' + ' | '.join(f'pad {p:3d}: {t:6.2f} us' for p, t in res.items()), flush=True)
