"""GPU parity of the stand-alone operators, called through the C ABI (libexorl_hip.so) and checked
against the oracle / fp64 numpy on the same seeded inputs."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def lib():
    from exorl_amd import _lib
    return _lib.load()


def dev(x):
    return torch.as_tensor(np.ascontiguousarray(x)).cuda()


def bf16_round(x):
    return torch.from_numpy(x).to(torch.bfloat16).to(torch.float32).numpy()


SHAPES = [(64, 64, 32), (100, 70, 50), (1, 6, 1024), (1024, 1024, 1024), (1024, 30, 1024), (37, 129, 67), (2048, 1024, 24)]


@pytest.mark.parametrize('prec', [0, 1, 2, 3])
@pytest.mark.parametrize('al,bl', [(0, 0), (0, 1), (1, 1), (1, 0)])
@pytest.mark.parametrize('M,N,K', SHAPES)
def test_gemm(lib, prec, al, bl, M, N, K):
    from exorl_amd import _lib as L
    rs = np.random.RandomState(M * 7 + N * 3 + K + al * 2 + bl)
    A = rs.standard_normal((M, K)).astype(np.float32)
    B = rs.standard_normal((K, N)).astype(np.float32)         # logical [K][N]
    bias = rs.standard_normal(N).astype(np.float32)
    C0 = rs.standard_normal((M, N)).astype(np.float32)
    A_st = A if al == 0 else np.ascontiguousarray(A.T)         # al=1: stored [K][M]
    B_st = np.ascontiguousarray(B.T) if bl == 0 else B         # bl=0: stored [N][K]
    a, b, c, bi = dev(A_st), dev(B_st), dev(C0.copy()), dev(bias)
    for relu, acc in ((0, 0), (1, 0), (0, 1)):
        c.copy_(torch.from_numpy(C0))
        L.check(lib.exorl_gemm(prec, al, bl, M, N, K, a.data_ptr(), A_st.shape[1], b.data_ptr(), B_st.shape[1],
                               c.data_ptr(), N, bi.data_ptr(), relu, acc, None))
        torch.cuda.synchronize()
        Ar, Br = (bf16_round(A), bf16_round(B)) if prec == 1 else (A, B)
        ref = Ar.astype(np.float64) @ Br.astype(np.float64) + bias
        if relu:
            ref = np.maximum(ref, 0)
        if acc:
            ref = ref + C0
        scale = np.abs(Ar).astype(np.float64) @ np.abs(Br).astype(np.float64) + 1.0
        err = np.abs(c.cpu().numpy() - ref) / scale
        # fp32 accumulation order only; split-bf16 (prec 2) drops the lo*lo term and the third bf16 digit: <= 3 * 2^-18 per product;
        # three planes (prec 3) hold all 24 significand bits and drop three terms of <= 2^-24 each: the fp32 bar
        assert err.max() < (1.2e-5 if prec == 2 else 2e-6), (prec, al, bl, M, N, K, relu, acc, err.max())


@pytest.mark.parametrize('al,bl', [(0, 0), (0, 1), (1, 1)])
@pytest.mark.parametrize('M,N,K', [(256, 328, 384), (512, 1000, 256), (256, 39200 // 8, 256), (384, 300, 640)])
def test_gemm_split_bf16_adapter_ragged_width_in_place(lib, al, bl, M, N, K):
    """fp32-source split-bf16 GEMMs with M, N, K >= 256 run on the hi/lo-plane kernels through the planes adapter (gemm.hip). A width that is not
    a multiple of the 128-column tile (the pixel agents' 39200-wide module layers) is written IN PLACE with the tile columns past N skipped in the
    epilogue (Gemm16Problem::n_store) instead of through a padded copy: the result against the float64 product, bias / ReLU / accumulate, a row
    pitch wider than N whose guard columns must stay untouched, and bit-equality with the padded-copy path (exorl_gemm_tune bit 134217728).
    M = 384 is three row tiles; (384, 300, 640) has K not a multiple of 128 as well."""
    from exorl_amd import _lib as L
    rs = np.random.RandomState(M + 3 * N + K + 11 * al + bl)
    A = rs.standard_normal((M, K)).astype(np.float32)
    B = rs.standard_normal((K, N)).astype(np.float32)
    bias = rs.standard_normal(N).astype(np.float32)
    ldc = N + 8
    C0 = rs.standard_normal((M, ldc)).astype(np.float32)
    A_st = A if al == 0 else np.ascontiguousarray(A.T)
    B_st = np.ascontiguousarray(B.T) if bl == 0 else B
    a, b, c, bi = dev(A_st), dev(B_st), dev(C0.copy()), dev(bias)
    for relu, acc in ((0, 0), (1, 0), (0, 1)):
        outs = []
        for bits in (0, 134217728):
            c.copy_(torch.from_numpy(C0))
            try:
                lib.exorl_gemm_tune(bits if bits else -1)
                L.check(lib.exorl_gemm(2, al, bl, M, N, K, a.data_ptr(), A_st.shape[1], b.data_ptr(), B_st.shape[1], c.data_ptr(), ldc, bi.data_ptr(), relu, acc, None))
            finally:
                lib.exorl_gemm_tune(-1)
            torch.cuda.synchronize()
            outs.append(c.cpu().numpy().copy())
        got = outs[0]
        assert np.array_equal(got[:, N:], C0[:, N:]), 'guard columns written'
        assert np.array_equal(outs[0], outs[1]), 'in-place and padded-copy results differ'
        ref = A.astype(np.float64) @ B.astype(np.float64) + bias
        if relu:
            ref = np.maximum(ref, 0)
        if acc:
            ref = ref + C0[:, :N]
        scale = np.abs(A).astype(np.float64) @ np.abs(B).astype(np.float64) + 1.0
        err = np.abs(got[:, :N] - ref) / scale
        assert err.max() < 1.2e-5, (al, bl, M, N, K, relu, acc, err.max())


@pytest.mark.parametrize('al,bl', [(0, 0), (0, 1), (1, 1)])
@pytest.mark.parametrize('M,N,K', [(64, 64, 64), (128, 64, 32), (1024, 1024, 1024), (2048, 1024, 1024), (96, 200, 72), (8, 32, 8),
                                   (256, 384, 512), (128, 128, 256), (192, 128, 256)])
def test_gemm_bf16_operands(lib, al, bl, M, N, K):
    """Fast-mode kernel: operands already bf16 in memory; exact up to fp32 accumulation order."""
    from exorl_amd import _lib as L
    rs = np.random.RandomState(M + N + K + 5 * al + bl)
    A = bf16_round(rs.standard_normal((M, K)).astype(np.float32))
    B = bf16_round(rs.standard_normal((K, N)).astype(np.float32))
    B[0, :] += bf16_round(np.arange(N, dtype=np.float32) % 7)       # asymmetric
    B = bf16_round(B)
    bias = rs.standard_normal(N).astype(np.float32)
    C0 = rs.standard_normal((M, N)).astype(np.float32)
    A_st = A if al == 0 else np.ascontiguousarray(A.T)
    B_st = np.ascontiguousarray(B.T) if bl == 0 else B
    a = torch.from_numpy(A_st).cuda().to(torch.bfloat16).contiguous()
    b = torch.from_numpy(B_st).cuda().to(torch.bfloat16).contiguous()
    c, bi = dev(C0.copy()), dev(bias)
    for relu, acc in ((0, 0), (1, 0), (0, 1)):
        c.copy_(torch.from_numpy(C0))
        L.check(lib.exorl_gemm_bf16(al, bl, M, N, K, a.data_ptr(), A_st.shape[1], b.data_ptr(), B_st.shape[1], c.data_ptr(), N,
                                    bi.data_ptr(), relu, acc, None))
        torch.cuda.synchronize()
        ref = A.astype(np.float64) @ B.astype(np.float64) + bias
        if relu:
            ref = np.maximum(ref, 0)
        if acc:
            ref = ref + C0
        scale = np.abs(A).astype(np.float64) @ np.abs(B).astype(np.float64) + 1.0
        err = np.abs(c.cpu().numpy() - ref) / scale
        assert err.max() < 2e-6, (al, bl, M, N, K, relu, acc, err.max())


@pytest.mark.parametrize('al', [0, 1])
@pytest.mark.parametrize('M,N,K', [(128, 128, 256), (256, 384, 512), (1024, 1024, 1024)])
def test_gemm_plain_bf16_k_image_regression(lib, al, M, N, K):
    """Round 2's unexplained wrong results: plain-bf16 planes with a k-image B operand under the two-region stage refill (SPREAD = 2) came out
    wrong and run-to-run different — in 20 of 20 launches at 1024^3. Cause (round 3, read off the ISA with tools/check_async_reads.py): the fence
    of the asm-issued ds_read_b64_tr_b16 fragments sat behind a branch, and the register allocator resolved the join with v_mov copies of their
    destination registers BEFORE the wait. Fixed in gemm16p_body (fence in the issuing basic block); plain bf16 now takes SPREAD = 2 by default,
    so this is the formerly failing launch: 20 launches per shape, each equal to the float64 product, all bit-identical."""
    from exorl_amd import _lib as L
    rs = np.random.RandomState(0)
    A = bf16_round(rs.standard_normal((M, K)).astype(np.float32))
    B = bf16_round(rs.standard_normal((K, N)).astype(np.float32))
    bias = rs.standard_normal(N).astype(np.float32)
    A_st = A if al == 0 else np.ascontiguousarray(A.T)
    a = torch.from_numpy(A_st).cuda().to(torch.bfloat16).contiguous()
    b = torch.from_numpy(B).cuda().to(torch.bfloat16).contiguous()
    bi = dev(bias)
    ref = A.astype(np.float64) @ B.astype(np.float64) + bias
    scale = np.abs(A).astype(np.float64) @ np.abs(B).astype(np.float64) + 1.0
    first = None
    for it in range(20):
        c = torch.full((M, N), float('nan'), device='cuda')
        L.check(lib.exorl_gemm_bf16(al, 1, M, N, K, a.data_ptr(), A_st.shape[1], b.data_ptr(), N, c.data_ptr(), N, bi.data_ptr(), 0, 0, None))
        torch.cuda.synchronize()
        got = c.cpu().numpy()
        assert (np.abs(got - ref) / scale).max() < 2e-6, (al, M, N, K, it)
        first = got if first is None else first
        assert np.array_equal(got, first), (al, M, N, K, it)


def test_gemm_first_launch_of_a_fresh_process():
    """The single-tile form of the same failure showed only in the FIRST launch of a process (cold instruction cache: the waves of a workgroup
    drift apart, the LDS returns late). One fresh interpreter per layout pair makes exactly one launch of each kernel family — plain-bf16 and
    split-bf16 planes, k-image operands — and compares it with the float64 product."""
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    code = r"""
import sys, ctypes as C
import numpy as np, torch
sys.path.insert(0, %r)
from exorl_amd import _lib as L
lib = L.load()
al, x3 = int(sys.argv[1]), int(sys.argv[2])
M, N, K = 128, 128, 256
g = torch.Generator(device='cuda').manual_seed(1)
A = torch.randn((M, K) if al == 0 else (K, M), device='cuda', generator=g)
B = torch.randn((K, N), device='cuda', generator=g)
split = lambda x: (x.to(torch.bfloat16), (x - x.to(torch.bfloat16).float()).to(torch.bfloat16))
(ah, al_), (bh, bl_) = split(A), split(B)
c = torch.full((M, N), float('nan'), device='cuda')
if x3:
    arr = lambda t: (C.c_void_p * 1)(t.data_ptr())
    L.check(lib.exorl_gemm_planes(1, (C.c_int32 * 1)(al), 1, M, N, K, arr(ah), arr(al_), A.shape[1], arr(bh), arr(bl_), N, arr(c), N, 0, None))
    Ad, Bd = ah.double() + al_.double(), bh.double() + bl_.double()
    lolo = al_.double() @ bl_.double() if al == 0 else al_.double().T @ bl_.double()
else:
    L.check(lib.exorl_gemm_bf16(al, 1, M, N, K, ah.data_ptr(), A.shape[1], bh.data_ptr(), N, c.data_ptr(), N, None, 0, 0, None))
    Ad, Bd, lolo = ah.double(), bh.double(), 0.0
torch.cuda.synchronize()
ref = (Ad if al == 0 else Ad.T) @ Bd - lolo
scale = (Ad.abs() if al == 0 else Ad.abs().T) @ Bd.abs() + 1.0
err = float(((c.double() - ref).abs() / scale).max())
print('ERR', err)
sys.exit(0 if err < 2e-6 else 3)
""" % str(root)
    for al in (0, 1):
        for x3 in (0, 1):
            r = subprocess.run([sys.executable, '-c', code, str(al), str(x3)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
            assert r.returncode == 0, (al, x3, r.stdout[-2000:])


def test_gemm_three_plane_products_are_fp32_grade(lib):
    """EXORL_PREC_BF16X6: hi + mid + lo hold an fp32 value exactly, six of the nine plane products are formed. On operands whose products need
    more than 16 bits — integers up to 2^11 against integers up to 2^11 — split-bf16 (prec 2) is visibly inexact, three planes are exact."""
    from exorl_amd import _lib as L
    rs = np.random.RandomState(0)
    A = rs.randint(-2048, 2048, (64, 64)).astype(np.float32)
    B = rs.randint(-2048, 2048, (64, 64)).astype(np.float32)
    a, b = dev(A), dev(np.ascontiguousarray(B.T))
    want = A.astype(np.float64) @ B.astype(np.float64)
    err = {}
    for prec in (2, 3):
        c = torch.zeros(64, 64, device='cuda')
        L.check(lib.exorl_gemm(prec, 0, 0, 64, 64, 64, a.data_ptr(), 64, b.data_ptr(), 64, c.data_ptr(), 64, None, 0, 0, None))
        err[prec] = float(np.abs(c.cpu().numpy().astype(np.float64) - want).max() / np.abs(want).max())
    assert err[3] < 2e-7 and err[2] > 10 * err[3] + 1e-7, err


def test_gemm_f32_is_exact_fp32_products(lib):
    """Parity mode must not round operands: integers up to 2^12 multiply exactly in fp32 MFMA."""
    from exorl_amd import _lib as L
    rs = np.random.RandomState(0)
    A = rs.randint(-4096, 4096, (64, 64)).astype(np.float32)
    B = rs.randint(-4, 4, (64, 64)).astype(np.float32)
    B[np.arange(64), np.arange(64)] += 7              # asymmetric
    a, b, c = dev(A), dev(np.ascontiguousarray(B.T)), torch.zeros(64, 64, device='cuda')
    L.check(lib.exorl_gemm(0, 0, 0, 64, 64, 64, a.data_ptr(), 64, b.data_ptr(), 64, c.data_ptr(), 64, None, 0, 0, None))
    assert np.array_equal(c.cpu().numpy(), (A.astype(np.float64) @ B.astype(np.float64)).astype(np.float32))


@pytest.mark.parametrize('rows,H', [(8, 32), (5, 50), (1024, 1024), (3, 1000)])
def test_ln_tanh_fwd(lib, rows, H):
    from exorl_amd import _lib as L
    from oracle import nets
    rs = np.random.RandomState(rows + H)
    z = (rs.standard_normal((rows, H)) * 2 + 0.3).astype(np.float32)
    g = (1 + 0.1 * rs.standard_normal(H)).astype(np.float32)
    b = (0.1 * rs.standard_normal(H)).astype(np.float32)
    zd, gd, bd = dev(z), dev(g), dev(b)
    h, xh, rstd = torch.empty_like(zd), torch.empty_like(zd), torch.empty(rows, device='cuda')
    L.check(lib.exorl_ln_tanh_fwd(zd.data_ptr(), gd.data_ptr(), bd.data_ptr(), h.data_ptr(), xh.data_ptr(), rstd.data_ptr(),
                                  rows, H, None))
    y, xhat, rs_ = nets.layernorm_fwd(z, g, b)
    np.testing.assert_allclose(xh.cpu().numpy(), xhat, rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(h.cpu().numpy(), np.tanh(y), rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(rstd.cpu().numpy(), rs_[:, 0], rtol=2e-6)
    # in place (z aliases h) is how the agent calls it
    L.check(lib.exorl_ln_tanh_fwd(zd.data_ptr(), gd.data_ptr(), bd.data_ptr(), zd.data_ptr(), None, None, rows, H, None))
    assert torch.equal(zd, h)


def test_adam_and_soft_update_golden(lib, gold):
    """Against the reference's own torch.optim.Adam / soft_update_params outputs (utils_g2.npz)."""
    from exorl_amd import _lib as L
    z = np.load(gold / 'utils_g2.npz')
    n = 36                                              # 33 padded to a multiple of 4
    p = torch.zeros(n, device='cuda'); p[:33] = dev(z['adam_p0'])
    m, v, g = torch.zeros(n, device='cuda'), torch.zeros(n, device='cuda'), torch.zeros(n, device='cuda')
    for t, (grad, want) in enumerate(zip(z['adam_grads'], z['adam_p']), 1):
        g[:33] = dev(grad)
        L.check(lib.exorl_adam_step(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), n, 1e-4, 0.9, 0.999, 1e-8, t,
                                    None, 0.0, None))
        np.testing.assert_allclose(p[:33].cpu().numpy(), want, rtol=0, atol=1e-9)
    assert float(p[33:].abs().max()) == 0.0             # padding stays zero
    w, tw = dev(z['soft_w']).reshape(-1), dev(z['soft_tw0']).reshape(-1)
    L.check(lib.exorl_soft_update(w.data_ptr(), tw.data_ptr(), w.numel(), 0.01, None))
    np.testing.assert_allclose(tw.cpu().numpy(), z['soft_tw1'].reshape(-1), rtol=0, atol=6e-8)


def test_adam_fused_soft_update_equals_separate(lib):
    from exorl_amd import _lib as L
    rs = np.random.RandomState(3)
    n = 4096
    p0, g0, t0 = [rs.standard_normal(n).astype(np.float32) for _ in range(3)]
    pa, ga, ma, va, ta = dev(p0), dev(g0), torch.zeros(n, device='cuda'), torch.zeros(n, device='cuda'), dev(t0)
    pb, mb, vb, tb = dev(p0), torch.zeros(n, device='cuda'), torch.zeros(n, device='cuda'), dev(t0)
    L.check(lib.exorl_adam_step(pa.data_ptr(), ga.data_ptr(), ma.data_ptr(), va.data_ptr(), n, 1e-4, 0.9, 0.999, 1e-8, 1,
                                ta.data_ptr(), 0.01, None))
    L.check(lib.exorl_adam_step(pb.data_ptr(), ga.data_ptr(), mb.data_ptr(), vb.data_ptr(), n, 1e-4, 0.9, 0.999, 1e-8, 1, None, 0.0, None))
    L.check(lib.exorl_soft_update(pb.data_ptr(), tb.data_ptr(), n, 0.01, None))
    assert torch.equal(pa, pb) and torch.equal(ta, tb)


@pytest.mark.parametrize('ns,nt,dim,k', [(16, 16, 8, 3), (10, 20, 6, 3), (1024, 1024, 512, 12), (1024, 2048, 128, 3), (7, 100, 70, 5)])
def test_knn_topk(lib, gold, ns, nt, dim, k):
    from exorl_amd import _lib as L
    from oracle import knn
    rs = np.random.RandomState(ns + nt)
    same = ns == nt
    src = rs.standard_normal((ns, dim)).astype(np.float32)
    tgt = src if same else rs.standard_normal((nt, dim)).astype(np.float32)
    out = torch.empty(ns, k, device='cuda')
    sd, td = dev(src), dev(tgt)
    L.check(lib.exorl_knn_topk(sd.data_ptr(), ns, td.data_ptr(), nt, dim, k, out.data_ptr(), None))
    torch.cuda.synchronize()
    want = knn.topk_smallest(knn.pairwise_l2(src[:64], tgt), k) if ns > 64 else knn.topk_smallest(knn.pairwise_l2(src, tgt), k)
    got = out.cpu().numpy()[:want.shape[0]]
    np.testing.assert_allclose(got, want, rtol=2e-5, atol=1e-6)
    if same:
        assert np.all(got[:, 0] == 0.0)            # self-distance is an exact zero (PBE relies on it)
    assert np.all(np.diff(out.cpu().numpy(), axis=1) >= 0)


def test_knn_golden_pbe_and_proto(lib, gold):
    """The reference's PBE / Proto rewards (utils_g2.npz) rebuilt from the HIP top-k."""
    from exorl_amd import _lib as L
    z = np.load(gold / 'utils_g2.npz')
    rep = z['pbe_rep']
    out = torch.empty(16, 3, device='cuda')
    rd, zd, qd = dev(rep), dev(z['knn_z']), dev(z['knn_queue'])
    L.check(lib.exorl_knn_topk(rd.data_ptr(), 16, rd.data_ptr(), 16, 8, 3, out.data_ptr(), None))
    topk = out.cpu().numpy()
    np.testing.assert_allclose(np.log(np.maximum(topk, 0).mean(1, keepdims=True) + 1), z['pbe_avg_r1'], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(np.log(topk[:, -1:] + 1), z['pbe_kth_r1'], rtol=1e-5, atol=1e-6)
    out = torch.empty(10, 3, device='cuda')
    L.check(lib.exorl_knn_topk(zd.data_ptr(), 10, qd.data_ptr(), 20, 6, 3, out.data_ptr(), None))
    np.testing.assert_allclose(out.cpu().numpy()[:, -1:], z['knn_reward'], rtol=1e-5)


@pytest.mark.parametrize('x3', [True, False])
@pytest.mark.parametrize('count,a_layouts,bl,M,N,K', [
    (4, [0, 0, 0, 0], 0, 1024, 1024, 1024),      # critic + target forward
    (4, [1, 1, 0, 0], 1, 1024, 1024, 1024),      # critic wgrad + dgrad in one launch
    (2, [0, 0], 0, 1024, 1024, 1024), (2, [0, 0], 1, 1024, 1024, 1024), (2, [1, 0], 1, 1024, 1024, 1024), (1, [0], 0, 2048, 1024, 1024),
    (1, [0], 1, 128, 128, 256), (1, [0], 1, 128, 64, 512), (1, [1], 1, 256, 384, 512), (2, [0, 0], 0, 128, 128, 128), (1, [0], 0, 128, 192, 384),
    (4, [0, 0, 0, 0], 1, 512, 512, 256), (2, [1, 1], 1, 256, 128, 640),
    (2, [0, 0], 0, 10240, 1024, 1024), (2, [0, 0], 1, 4096, 1024, 1024), (2, [1, 1], 1, 1024, 1024, 4096)])       # CQL's 10 B-row pass; B = 4096: dgrad rows, wgrad reduction length
def test_gemm_planes_shapes(lib, x3, count, a_layouts, bl, M, N, K):
    """The grouped H x H GEMM on bf16 hi/lo planes (exorl_gemm_planes: what the agent's six launches call), every operand layout, launch shapes
    from one stage ring (K = 128) to 32, with REAL lo planes: equal to the float64 product of the same planes (minus lo*lo, which the kernel
    drops) to fp32 accumulation order, and bit-identical over repeated launches (a stage-ring race shows up as a result that moves)."""
    from exorl_amd import _lib as L
    C = L.C
    g = torch.Generator(device='cuda').manual_seed(M + N + K + count + bl)

    def split(x):
        hi = x.to(torch.bfloat16)
        return hi, (x - hi.float()).to(torch.bfloat16)
    ps = []
    for i in range(count):
        A = torch.randn((M, K) if a_layouts[i] == 0 else (K, M), device='cuda', generator=g)
        B = torch.randn((N, K) if bl == 0 else (K, N), device='cuda', generator=g)
        ps.append((split(A), split(B), torch.zeros(M, N, device='cuda')))
    arr = lambda xs: (C.c_void_p * count)(*[x.data_ptr() for x in xs])
    lay = (C.c_int32 * count)(*a_layouts)
    lda, ldb = (K if a_layouts[0] == 0 else M), (K if bl == 0 else N)

    def launch():
        L.check(lib.exorl_gemm_planes(count, lay, bl, M, N, K, arr([p[0][0] for p in ps]), arr([p[0][1] for p in ps]) if x3 else None, lda,
                                      arr([p[1][0] for p in ps]), arr([p[1][1] for p in ps]) if x3 else None, ldb, arr([p[2] for p in ps]), N, 0,
                                      torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        return [p[2].clone() for p in ps]
    first = launch()
    for i, ((ah, alo), (bh, blo), _) in enumerate(ps):
        Ad = ah.double() + (alo.double() if x3 else 0)
        Bd = bh.double() + (blo.double() if x3 else 0)
        Ad = Ad if a_layouts[i] == 0 else Ad.t()
        Bd = Bd.t() if bl == 0 else Bd
        ref = Ad @ Bd
        if x3:
            ref = ref - (alo.double() if a_layouts[i] == 0 else alo.double().t()) @ (blo.double().t() if bl == 0 else blo.double())
        err = float((first[i].double() - ref).abs().max() / ref.abs().max())
        assert err < 2e-6, (i, err)
    for _ in range(6):
        again = launch()
        assert all(torch.equal(a, b) for a, b in zip(first, again))
