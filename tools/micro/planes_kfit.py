"""Fixed cost vs per-k cost of a grouped launch: time at K = 128 .. 1024 (same outputs, same launch shape)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent))
from planes_bench import make, timed, H
import numpy as np
for x3 in (True, False):
    for tag, count, lay, bl in (('fwd x4', 4, [0, 0, 0, 0], 0), ('fwd x2', 2, [0, 0], 0), ('dgrad x2', 2, [0, 0], 1)):
        for variant, vn in ((0, 'new'), (262144, 'old')):
            ks, ts = [128, 256, 512, 1024], []
            for K in ks:
                ps = make(count, lay, bl, 1024, H, K, x3)
                ts.append(timed(ps, lay, bl, 1024, H, K, x3, variant))
            slope, icpt = np.polyfit(ks, ts, 1)
            print(f'{"x3" if x3 else "bf16":5s} {tag:9s} {vn}: ' + ' '.join(f'K={k}: {t:6.2f}' for k, t in zip(ks, ts)) +
                  f' | fixed {icpt:5.2f} us (incl. ~4.4 us event bracket), {slope * 1024:6.2f} us per 1024 k', flush=True)
