"""A/B of the MFMA shape in the forward (row x row image) launches of the H x H GEMM: 32x32x16 (default) against 16x16x32 (exorl_gemm_tune
bit 4194304; needs a library built with -DEXORL_GEMM_EXPERIMENTS), same tiles / stages / LDS images; correctness against a float64 product of the same planes, then interleaved timing.
python tools/micro/ms16_bench.py"""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
sys.path.insert(0, str(Path(__file__).resolve().parent))
from planes_bench import make, launch, reference, timed, lib, H

SHAPES = [
    ('critic+target fwd (4)', 4, [0, 0, 0, 0], 0, 1024, H, H),
    ('critic fwd (2)', 2, [0, 0], 0, 1024, H, H),
    ('actor fwd 2B (1)', 1, [0], 0, 2048, H, H),
]
MS16 = 4194304
for x3 in (True,):
    for tag, count, lay, bl, M, N, K in SHAPES:
        ps = make(count, lay, bl, M, N, K, x3)
        for variant in (0, MS16, MS16 | 524288):
            lib.exorl_gemm_tune(variant)
            for p in ps:
                p[2].zero_()
            launch(ps, lay, bl, M, N, K, x3, relu=0)
            torch.cuda.synchronize()
            worst = 0.0
            for i, p in enumerate(ps):
                ref = reference(p, lay[i], bl, x3)
                worst = max(worst, float((p[2].double() - ref).abs().max() / ref.abs().max()))
            assert worst < 2e-6, (tag, x3, variant, worst)
        lib.exorl_gemm_tune(-1)
        rounds = [{v: timed(ps, lay, bl, M, N, K, x3, v, iters=200) for v in (0, MS16, 524288, MS16 | 524288)} for _ in range(3)]
        fl = 2.0 * M * N * K * count
        for r in rounds:
            print(f'{tag:24s} 32x32x16 {r[0]:6.2f} us ({fl / r[0] / 1e6:4.0f} TF/s) | 16x16x32 {r[MS16]:6.2f} us ({fl / r[MS16] / 1e6:4.0f} TF/s) | '
                  f'128x128 everywhere: {r[524288]:6.2f} vs {r[MS16 | 524288]:6.2f}   (err {worst:.1e})', flush=True)
