"""One grouped split-bf16 launch shape, repeated, for rocprofv3 --pmc passes: python planes_one.py <variant> [count]"""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent))
from planes_bench import make, launch, lib, H
variant = int(sys.argv[1])
count = int(sys.argv[2]) if len(sys.argv) > 2 else 4
ps = make(count, [0] * count, 0, 1024, H, H, True)
lib.exorl_gemm_tune(variant)
for _ in range(20):
    launch(ps, [0] * count, 0, 1024, H, H, True)
torch.cuda.synchronize()
