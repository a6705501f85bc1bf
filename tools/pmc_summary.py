"""Aggregates a rocprofv3 --pmc counter_collection CSV per kernel (mean per dispatch)."""
import csv
import sys
from collections import defaultdict

path = sys.argv[1]
agg = defaultdict(lambda: defaultdict(list))
for r in csv.DictReader(open(path)):
    n = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('exorl::', '')[:44]
    agg[n][r['Counter_Name']].append(float(r['Counter_Value']))
names = sorted({c for k in agg.values() for c in k})
print(f'{"kernel":44s} {"disp":>6s} ' + ' '.join(f'{c[-18:]:>18s}' for c in names))
for k, d in sorted(agg.items(), key=lambda kv: -sum(kv[1].get(names[0], [0]))):
    nd = max(len(v) for v in d.values())
    print(f'{k:44s} {nd:6d} ' + ' '.join(f'{(sum(d[c]) / len(d[c]) if c in d else 0):18.1f}' for c in names))
