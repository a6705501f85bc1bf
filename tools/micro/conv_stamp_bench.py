"""Phase split of the 32 -> 32 forward convolution at config-4 shapes (1024 images, 41 -> 39 -> 37 -> 35): the stamped diagnostic build
(exorl_gemm_tune bit 8192) of conv3x3_mfma_kernel runs inside the encoder forward; the stamps read back are those of the LAST stamped
launch, layer 4 (37 -> 35: 5 passes per image against layer 2's 6, same strip width class).     python tools/micro/conv_stamp_bench.py [bf16x3|bf16x6]"""
import ctypes as C
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from exorl_amd import _lib as L

lib = L.load()
prec = {'bf16x3': L.PREC_BF16X3, 'bf16x6': L.PREC_BF16X6}[sys.argv[1] if len(sys.argv) > 1 else 'bf16x6']
STRIP = 1073741824 if '--strip' in sys.argv else 0
n, c_in, hw = 1024, 9, 84
rs = np.random.RandomState(0)
P = torch.from_numpy((rs.randn(lib.exorl_encoder_param_floats(c_in, hw)) * 0.05).astype(np.float32)).cuda()
X = torch.from_numpy(rs.uniform(0, 255, (n, c_in, hw, hw)).astype(np.float32)).cuda()
ws = torch.zeros(lib.exorl_encoder_workspace_floats(n, c_in, hw), device='cuda')
hp = C.c_void_p()
for stamp in (0, 8192):
    lib.exorl_gemm_tune((stamp | STRIP) if (stamp | STRIP) else -1)
    for _ in range(3):
        L.check(lib.exorl_encoder_forward_prec(P.data_ptr(), c_in, hw, X.data_ptr(), n, ws.data_ptr(), C.byref(hp), prec, None))
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(20):
        L.check(lib.exorl_encoder_forward_prec(P.data_ptr(), c_in, hw, X.data_ptr(), n, ws.data_ptr(), C.byref(hp), prec, None))
    torch.cuda.synchronize()
    print(f'encoder forward, {"stamped" if stamp else "product"} build: {(time.time() - t0) / 20 * 1e3:.3f} ms', flush=True)
buf = np.zeros(1024 * 16, np.uint64)
L.check(lib.exorl_debug_conv_stamps(buf.ctypes.data, buf.size))
lib.exorl_gemm_tune(-1)
st = buf.reshape(1024, 2, 8).astype(np.int64)
if '--strip' in sys.argv:
    names = ['convert+LDS write', 'barrier 1', 'fetch issue', 'MFMA loop', 'stores', 'barrier 2', 'whole kernel']
    npass = (35 * 35 + 255) // 256
    mf = 18 * (6 if prec == L.PREC_BF16X6 else 3) * 32 * npass
    print(f'strip kernel, last layer (37 -> 35, {npass} passes per image); MFMA issue floor per wave {mf} cycles (x2 waves per SIMD)')
    for w, tag in ((0, 'wave 0'), (1, 'wave 7')):
        print(f'{tag}: ' + ' | '.join(f'{nm} {np.median(st[:, w, i]):7.0f}' for i, nm in enumerate(names)))
else:
    npass = (35 * 35 + 127) // 128
    mf = 18 * (6 if prec == L.PREC_BF16X6 else 3) * 32 * npass
    print(f'wave-specialised kernel (persistent form: per-image averages), last layer (37 -> 35, {npass} passes per image); MFMA issue floor of a consumer wave {mf} cycles')
    per_wg = max(1, -(-n // 256))             # persistent form: a workgroup walks n / 256 images and sums its stamps over them
    st = st[:min(n, 256)] // per_wg
    c, p_ = st[:, 0], st[:, 1]
    print(f'consumer wave 0: k-loops (with the previous pass leaving) {np.median(c[:, 0]):7.0f} | barrier wait + next addresses {np.median(c[:, 1]):7.0f} | prologue {np.median(c[:, 7]):6.0f} | whole kernel {np.median(c[:, 6]):7.0f}')
    print(f'producer wave 4 (even passes: commit | fetch issue | barrier; odd passes: commit + fetch | barrier): {np.median(p_[:, 0]):7.0f} | {np.median(p_[:, 1]):7.0f} | '
          f'{np.median(p_[:, 2]):7.0f} ; {np.median(p_[:, 3]):7.0f} | {np.median(p_[:, 4]):7.0f} | prologue {np.median(p_[:, 7]):6.0f} | whole kernel {np.median(p_[:, 6]):7.0f} | of the even commits, waiting for the batch: {np.median(p_[:, 5]):7.0f}')
