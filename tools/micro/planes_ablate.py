"""Where the split-bf16 128 x 128 kernel's time goes: the 4-problem forward launch with parts of the k-loop switched off."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent))
from planes_bench import make, timed, H
ps = make(4, [0, 0, 0, 0], 0, 1024, H, H, True)
for name, v in (('full', 0), ('no MFMA', 1 << 22), ('no DMA after prologue', 2 << 22), ('no fragment reads', 3 << 22), ('old kernel', 262144)):
    print(f'{name:24s} {timed(ps, [0, 0, 0, 0], 0, 1024, H, H, True, v):7.2f} us', flush=True)
ps = make(1, [0], 0, 128, 128, H, True)
print(f'one 128x128 tile alone   {timed(ps, [0], 0, 128, 128, H, True, 524288):7.2f} us')
ps = make(4, [0, 0, 0, 0], 0, 256, 512, H, True)
print(f'32 tiles (1 XCD-full)    {timed(ps, [0, 0, 0, 0], 0, 256, 512, H, True, 524288):7.2f} us')
