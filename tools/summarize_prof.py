"""Condense rocprofv3 csv output under gpurun_out/ into a small text summary: per-kernel launches / total / average
duration from the kernel trace, and per-kernel mean FETCH_SIZE / WRITE_SIZE from the PMC passes (units as reported by
rocprofv3; the gfx950 correction of MI355X_MICROARCH.md -- FETCH_SIZE reads 1/2 of a wide coalesced stream -- is applied
in profiles/README.md, not here). Usage: python tools/summarize_prof.py gpurun_out"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

out = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out'


def short(name):
    name = re.sub(r'\(.*$', '', name)
    return name[:110]


for f in sorted(glob.glob(os.path.join(out, 'prof', '**', '*kernel_stats.csv'), recursive=True)):
    print('== kernel stats', f)
    with open(f) as fh:
        for i, row in enumerate(csv.reader(fh)):
            if i < 30:
                print('  ' + ' | '.join(c[:90] for c in row))
for f in sorted(glob.glob(os.path.join(out, 'prof', '**', '*kernel_trace.csv'), recursive=True)):
    agg = defaultdict(lambda: [0, 0.0])
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = short(row.get('Kernel_Name', '?'))
            agg[k][0] += 1
            agg[k][1] += (int(row['End_Timestamp']) - int(row['Start_Timestamp'])) / 1e3
    print('== kernel trace', f)
    for k, (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
        print(f'  {us / 1e3:10.3f} ms total {n:7d} launches {us / n:10.2f} us avg  {k}')
for tag in ('pmc_fetch', 'pmc_write'):
    for f in sorted(glob.glob(os.path.join(out, tag, '**', '*counter_collection.csv'), recursive=True)):
        agg = defaultdict(lambda: [0, 0.0])
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = (short(row.get('Kernel_Name', '?')), row.get('Counter_Name', '?'))
                agg[k][0] += 1
                agg[k][1] += float(row.get('Counter_Value', 0))
        print('== pmc', f)
        for (k, c), (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
            print(f'  {c:12s} sum {v:16.1f} over {n:7d} dispatches, mean {v / n:14.2f}  {k}')
