"""Which tensors differ between the persistent wave-specialised convolution kernels, the one-image-per-workgroup form (bit 8388608) and the
round-2 kernels (bits 1073741824 | 64), per precision mode, at n = 300 images of 48 x 48."""
import sys, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch
from exorl_amd import _lib as L
lib = L.load()
import test_gpu_pixels as T
n, hw = int(sys.argv[1]) if len(sys.argv) > 1 else 300, 48
rs = np.random.RandomState(11)
p = []
for l in range(4):
    ci = 3 if l == 0 else 32
    p += [(rs.standard_normal((32, ci, 3, 3)) * np.sqrt(2.0 / (ci * 9))).astype(np.float32), (0.1 * rs.standard_normal(32)).astype(np.float32)]
x = rs.randint(0, 256, (n, 3, hw, hw)).astype(np.uint8)
edge = (hw - 3) // 2 + 1 - 6
dh = rs.standard_normal((n, 32 * edge * edge)).astype(np.float32)
def run(bits, prec):
    lib.exorl_gemm_tune(bits if bits else -1)
    try:
        return T.run_encoder(lib, p, x, dh, prec)
    finally:
        lib.exorl_gemm_tune(-1)
for prec in (2, 1, 3):
    ref = run(1073741824 | 64, prec)
    for tag, bits in (('persistent', 0), ('one image per wg', 8388608), ('persistent fwd/dgrad + tile wgrad', 64), ('strip fwd/dgrad + ws wgrad', 1073741824)):
        h, g = run(bits, prec)
        d = [float(np.abs(h - ref[0]).max() / np.abs(ref[0]).max())] + [float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30)) for a, b in zip(g, ref[1])]
        print(f'prec {prec} {tag:36s}: ' + ' '.join(f'{v:.1e}' for v in d), flush=True)
