"""Where a plain-bf16 GEMM launch (exorl_gemm_bf16) goes wrong: error by output tile and by which k-slabs explain it."""
import sys
from pathlib import Path
import numpy as np
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from exorl_amd import _lib as L
lib = L.load()
M, N, K = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (128, 128, 256)
al, bl = 0, 1
rs = np.random.RandomState(0)
A = rs.standard_normal((M, K)).astype(np.float32)
B = rs.standard_normal((K, N)).astype(np.float32)
a = torch.from_numpy(A).cuda().to(torch.bfloat16).contiguous()
b = torch.from_numpy(B).cuda().to(torch.bfloat16).contiguous()
Ab, Bb = a.float().cpu().numpy().astype(np.float64), b.float().cpu().numpy().astype(np.float64)
c = torch.zeros(M, N, device='cuda')
L.check(lib.exorl_gemm_bf16(al, bl, M, N, K, a.data_ptr(), K, b.data_ptr(), N, c.data_ptr(), N, None, 0, 0, None))
torch.cuda.synchronize()
got = c.cpu().numpy().astype(np.float64)
ref = Ab @ Bb
err = np.abs(got - ref)
print('max err', err.max(), 'mean', err.mean())
for r0 in range(0, M, 64):
    print(' '.join(f'{err[r0:r0 + 64, c0:c0 + 32].max():9.3g}' for c0 in range(0, N, 32)))
# which 32-wide k slabs, if dropped or doubled, explain the result of tile (0,0)?
d = (got - ref)[:64, :32]
for k0 in range(0, K, 32):
    slab = Ab[:64, k0:k0 + 32] @ Bb[k0:k0 + 32, :32]
    coef = float((d * slab).sum() / (slab * slab).sum())
    print(f'k slab {k0:4d}: coefficient {coef:+.3f}')

# the same product through the split-bf16 planes entry (lo planes zero -> same result), and repeated launches (is it stable?)
sys.path.insert(0, str(Path(__file__).resolve().parents[1] / 'micro'))
C = L.C
zero_a, zero_b = torch.zeros_like(a), torch.zeros_like(b)
c2 = torch.zeros(M, N, device='cuda')
arr = lambda t: (C.c_void_p * 1)(t.data_ptr())
lay = (C.c_int32 * 1)(0)
L.check(lib.exorl_gemm_planes(1, lay, 1, M, N, K, arr(a), arr(zero_a), K, arr(b), arr(zero_b), N, arr(c2), N, 0, None))
torch.cuda.synchronize()
print('planes (x3) max err', float(np.abs(c2.cpu().numpy() - ref).max()))
outs = []
for _ in range(5):
    c.zero_()
    L.check(lib.exorl_gemm_bf16(al, bl, M, N, K, a.data_ptr(), K, b.data_ptr(), N, c.data_ptr(), N, None, 0, 0, None))
    torch.cuda.synchronize()
    outs.append(c.cpu().numpy().copy())
print('plain bf16 repeat: identical results', all(np.array_equal(outs[0], o) for o in outs), 'max err', [float(np.abs(o - ref).max()) for o in outs])
