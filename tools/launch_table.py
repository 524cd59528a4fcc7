"""On-GPU probe: the per-launch table of one profiled 32-pair forward (kernel class, M, N, K, launches, ms, TFLOP/s), for one or more settings of an
environment switch. Usage: python tools/launch_table.py VAR=a,b [pairs]"""
import os
import sys

import torch

sys.path.insert(0, '.')
import bench  # noqa: E402

var, vals = sys.argv[1].split('=')
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
dev = torch.device('cuda:0')
model = bench.build_model(None, dev)
from dust3r_amd._lib import lib  # noqa: E402
from dust3r_amd.synthetic import synthetic_views  # noqa: E402
v1, v2 = synthetic_views(B, bench.H, bench.W, seed=0, device=dev)
tables = {}
for v in vals.split(','):
    os.environ[var] = v
    for _ in range(2):
        model(v1, v2)
    torch.cuda.synchronize()
    lib.d3r_model_set_option(model._engine, 1, 1)
    model(v1, v2)
    torch.cuda.synchronize()
    tables[v] = bench.read_launch_table(model)
    lib.d3r_model_set_option(model._engine, 1, 0)
keys = []
for v, t in tables.items():
    for r in t:
        k = (r['M'], r['N'], r['K'], 'attn' if r['kernel'] == 'attention' else 'conv' if r['kernel'].startswith('conv') else 'other' if r['kernel'] == 'other' else 'lin')
        if k not in keys:
            keys.append(k)
print(f'{"class":5s} {"M":>8s} {"N":>6s} {"K":>6s} | ' + ' | '.join(f'{var}={v}: kernel, launches, ms, TF/s' for v in tables))
tot = {v: 0.0 for v in tables}
for k in keys:
    cells = []
    for v, t in tables.items():
        rows = [r for r in t if (r['M'], r['N'], r['K']) == k[:3] and ('attn' if r['kernel'] == 'attention' else 'conv' if r['kernel'].startswith('conv') else 'other' if r['kernel'] == 'other' else 'lin') == k[3]]
        ms = sum(r['ms'] for r in rows)
        tot[v] += ms
        cells.append(', '.join(f"{r['kernel']} x{r['launches']} {r['ms']:.3f} {r['tflops']:.0f}" for r in rows) if rows else '-')
    print(f'{k[3]:5s} {k[0]:8d} {k[1]:6d} {k[2]:6d} | ' + ' | '.join(cells))
print('total ms (single-stream sum): ' + ', '.join(f'{v}: {t:.2f}' for v, t in tot.items()))
