"""Summarises a rocprofv3 --kernel-trace CSV: per-kernel totals and one steady-state step timeline."""
import csv
import sys
from collections import defaultdict

path, nsteps = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 1
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
agg = defaultdict(lambda: [0, 0])
for r in rows:
    n = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('exorl::', '')
    agg[n][0] += 1
    agg[n][1] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
tot = sum(v[1] for v in agg.values())
print(f'{"kernel":60s} {"calls":>7s} {"avg_us":>9s} {"us/step":>9s} {"%":>6s}')
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f'{n[:60]:60s} {c:7d} {t / c / 1e3:9.2f} {t / nsteps / 1e3:9.1f} {100 * t / tot:6.2f}')
print(f'total kernel time per step: {tot / nsteps / 1e3:.1f} us')
idx = [i for i, r in enumerate(rows) if 'step_begin' in r['Kernel_Name'] or 'gather_nstep' in r['Kernel_Name']]
if len(idx) > 20:
    a, b = idx[len(idx) // 2] - 1, idx[len(idx) // 2 + 1] - 1
    t0 = int(rows[a]['Start_Timestamp'])
    print('\none step:')
    for r in rows[a:b]:
        n = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('exorl::', '')[:50]
        print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:8.1f} {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:7.1f}  {n:50s} grid={int(r['Grid_Size_X'])//int(r['Workgroup_Size_X'])},{r['Grid_Size_Y']},{r['Grid_Size_Z']}")
