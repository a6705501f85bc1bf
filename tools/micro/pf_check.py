"""One launch of a forward GEMM shape under one exorl_gemm_tune mask, checked against the float64 product (experiments build): used to try a
new schedule once, on one shape, before it is timed.   python tools/micro/pf_check.py <mask> [count]"""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
sys.path.insert(0, str(Path(__file__).resolve().parent))
from planes_bench import make, launch, reference, lib, H
mask, count = int(sys.argv[1]), int(sys.argv[2]) if len(sys.argv) > 2 else 2
lay = [0] * count
ps = make(count, lay, 0, 1024, H, H, True)
lib.exorl_gemm_tune(mask)
launch(ps, lay, 0, 1024, H, H, True)
torch.cuda.synchronize()
worst = max(float((p[2].double() - reference(p, 0, 0, True)).abs().max() / reference(p, 0, 0, True).abs().max()) for p in ps)
print('mask', mask, 'count', count, 'max rel err', worst, flush=True)
assert worst < 2e-6
