"""From a rocprofv3 kernel trace: the generic gemm_kernel launches of the last update, by grid and duration (which of them are the 39200-wide layers)."""
import csv, sys
from collections import defaultdict
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'gemm_kernel' in r['Kernel_Name']]
per = int(sys.argv[2]) if len(sys.argv) > 2 else 13
n = len(rows) // per
last = rows[-n:]
agg = defaultdict(list)
for r in last:
    g = (r['Kernel_Name'].split('gemm_kernel')[1][:16], r.get('Grid_Size_X', r.get('Grid_Size', '?')), r.get('Grid_Size_Y', ''), r.get('Grid_Size_Z', ''))
    agg[g].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
tot = 0
for g, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print(f'{g[0]:18s} grid {g[1]:>7s} {g[2]:>4s} {g[3]:>3s}  x{len(v):3d}  avg {sum(v) / len(v):7.1f} us  total {sum(v):8.1f} us')
    tot += sum(v)
print('total', tot)
