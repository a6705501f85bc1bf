import sys, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch
from exorl_amd import _lib as L
lib = L.load()
import test_gpu_pixels as T
rs = np.random.RandomState(0)
for (n, c, hw) in ((5, 3, 64), (3, 9, 84)):
    p = []
    for l in range(4):
        ci = c if l == 0 else 32
        p += [(rs.standard_normal((32, ci, 3, 3)) / np.sqrt(ci * 9)).astype(np.float32), (0.1 * rs.standard_normal(32)).astype(np.float32)]
    x = rs.randint(0, 256, (n, c, hw, hw)).astype(np.uint8)
    dh = rs.standard_normal((n, 32 * ((hw - 3) // 2 + 1 - 6) ** 2)).astype(np.float32)
    for prec in (1, 2):
        for rep in range(3):
            h, g = T.run_encoder(lib, p, x, dh, prec)
            print(n, c, hw, 'prec', prec, 'rep', rep, 'nan per grad:', [int(np.isnan(a).sum()) for a in g], flush=True)
            for i in (2, 4, 6):
                a = g[i]
                if np.isnan(a).any():
                    idx = np.argwhere(np.isnan(a))
                    print('   grad', i, 'nan taps (ky,kx) counts:', {(int(ky), int(kx)): int(((idx[:, 2] == ky) & (idx[:, 3] == kx)).sum()) for ky in range(3) for kx in range(3)}, 'co range', idx[:, 0].min(), idx[:, 0].max(), 'ci range', idx[:, 1].min(), idx[:, 1].max())
