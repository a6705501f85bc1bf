"""A/B of schedule variants of the H x H GEMM on the agent's six launch shapes (split-bf16): correctness against a float64 product of the same
planes, then interleaved timing. Variants are exorl_gemm_tune masks: 0 = default (a stage's LDS-DMA refill spread over two k-regions),
536870912 = the previous schedule (whole refill in the region behind the barrier); in a build with -DEXORL_GEMM_EXPERIMENTS also
8388608 = wave-specialised 512-thread workgroups (4 MFMA waves + 4 loader waves), 4194304 = 16x16x32 MFMAs (forward launches),
1073741824 (+4194304) = per-wave DMA placement in 2 (4) phases, 268435456 = 512-thread workgroups whose 8 waves all load and compute
(forward launches).
    python tools/micro/ws_bench.py [more masks]"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
sys.path.insert(0, str(Path(__file__).resolve().parent))
from planes_bench import make, launch, reference, timed, lib, SHAPES

VARIANTS = [0, 536870912] + [int(v) for v in sys.argv[1:]]
for tag, count, lay, bl, M, N, K in SHAPES:
    ps = make(count, lay, bl, M, N, K, True)
    for variant in VARIANTS:
        lib.exorl_gemm_tune(variant)
        for p in ps:
            p[2].zero_()
        launch(ps, lay, bl, M, N, K, True)
        torch.cuda.synchronize()
        worst = 0.0
        for i, p in enumerate(ps):
            ref = reference(p, lay[i], bl, True)
            worst = max(worst, float((p[2].double() - ref).abs().max() / ref.abs().max()))
        assert worst < 2e-6, (tag, variant, worst)
    lib.exorl_gemm_tune(-1)
    rounds = [{v: timed(ps, lay, bl, M, N, K, True, v, iters=200) for v in VARIANTS} for _ in range(3)]
    best = {v: min(r[v] for r in rounds) for v in VARIANTS}
    fl = 2.0 * M * N * K * count
    print(f'{tag:26s} ' + ' | '.join(f'{v:>9d}: {best[v]:6.2f} us ({fl / best[v] / 1e6:4.0f} TF/s)' for v in VARIANTS), flush=True)
