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

# ---- SQ / GRBM passes: per kernel, mean per dispatch of every counter + the derived utilisations ------------------------------
for tag in ('pmc_sq1', 'pmc_sq2'):
    for f in sorted(glob.glob(os.path.join(out, tag, '**', '*counter_collection.csv'), recursive=True)):
        agg = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = short(row.get('Kernel_Name', '?'))
                if 'd3r::' not in k:
                    continue
                a = agg[k][row.get('Counter_Name', '?')]
                a[0] += 1
                a[1] += float(row.get('Counter_Value', 0))
        print('== pmc (SQ / GRBM), mean per dispatch:', f)
        order = sorted(agg.items(), key=lambda kv: -kv[1].get('GRBM_GUI_ACTIVE', [0, 0.0])[1])
        for k, cs in order[:14]:
            n = max(v[0] for v in cs.values())
            m = {c: v[1] / max(v[0], 1) for c, v in cs.items()}
            line = f'  {k[:100]}  x{n}\n      ' + '  '.join(f'{c}={v:.4g}' for c, v in sorted(m.items()))
            gui = m.get('GRBM_GUI_ACTIVE', 0.0)
            if gui > 0 and 'SQ_VALU_MFMA_BUSY_CYCLES' in m:
                # the csv row of GRBM_GUI_ACTIVE is the sum over the 8 XCDs (each has its own GRBM): 297 us launches read 5.4e6 "cycles"
                line += f'\n      MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs) = {m["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui / 8 * 1024) * 100:.1f} %'
            if m.get('SQ_WAVE_CYCLES', 0) > 0:
                w = m['SQ_WAVE_CYCLES']
                parts = [f'{c}/WAVE_CYCLES={m[c] / w * 100:.1f}%' for c in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_WAIT_INST_LDS', 'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_LDS', 'SQ_INST_CYCLES_VMEM') if c in m]
                line += '\n      ' + '  '.join(parts)
            print(line)

# ---- machine-readable digest for bench.py's roofline.traffic (copied by hand to profiles/pmc_latest.json) ------------------
import json  # noqa: E402

CFG_OF = {(1, 8, 12, 3, 2, 128, 2, 0): 9, (2, 2, 2, 2, 4, 128, 3, 0): 8, (2, 2, 2, 3, 4, 128, 3, 0): 8, (2, 4, 4, 2, 4, 128, 2, 0): 0, (2, 2, 4, 4, 2, 128, 2, 0): 0, (2, 4, 8, 4, 2, 128, 2, 0): 1, (2, 4, 4, 4, 2, 128, 2, 0): 2, (1, 8, 8, 4, 2, 128, 2, 0): 3,
          (1, 4, 8, 4, 2, 64, 3, 0): 4, (2, 4, 8, 4, 2, 64, 4, 0): 5,
          (2, 2, 4, 4, 2, 128, 2): 0, (2, 4, 8, 4, 2, 128, 2): 1, (2, 4, 4, 4, 2, 128, 2): 2, (1, 8, 8, 4, 2, 128, 2): 3,
          (1, 4, 8, 4, 2, 64, 3): 4, (2, 4, 8, 4, 2, 64, 4): 5, (2, 4, 8, 4, 2, 64, 4, 1): 6,
          # fp16 + fp8 rows, DMA pieces interleaved with the MFMA rows (PP = 4)
          (2, 2, 4, 4, 2, 128, 2, 4): 0, (2, 4, 8, 4, 2, 128, 2, 4): 1, (2, 4, 4, 4, 2, 128, 2, 4): 2, (1, 8, 8, 4, 2, 128, 2, 4): 3,
          (1, 4, 8, 4, 2, 128, 2, 6): 7}


def pmc_means(tag, counter):
    res = {}
    for f in sorted(glob.glob(os.path.join(out, tag, '**', '*counter_collection.csv'), recursive=True)):
        agg = defaultdict(lambda: [0, 0.0])
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row.get('Counter_Name') != counter:
                    continue
                agg[row.get('Kernel_Name', '?')][0] += 1
                agg[row.get('Kernel_Name', '?')][1] += float(row.get('Counter_Value', 0))
        for k, (n, v) in agg.items():
            res[k] = (n, v / n)
    return res


fetch, write = pmc_means('pmc_fetch', 'FETCH_SIZE'), pmc_means('pmc_write', 'WRITE_SIZE')
digest = {'note': 'per-launch means from separate rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of `bench.py --single-stream`; counters in KiB; '
                  'hbm_gb_per_launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 / 1e9 (FETCH_SIZE reads 1/2 of a wide coalesced stream on gfx950, '
                  'MI355X_MICROARCH.md section HBM; WRITE_SIZE uncalibrated)', 'gemm_cfg': {}}
DT_NAMES = {0: 'bf16', 1: 'fp16', 2: 'fp32', 3: 'fp16x3', 4: 'fp16f8'}
names = set(fetch) | set(write)
# the precision mode of the run = the element type of the most-dispatched gemm_kernel instantiation
dt_count = defaultdict(int)
for name in names:
    mm = re.search(r'gemm_kernel<(\d), d3r::GemmCfg<', name)
    if mm:
        dt_count[int(mm.group(1))] += fetch.get(name, write.get(name))[0]
run_dt = max(dt_count, key=dt_count.get) if dt_count else 0
digest['precision'] = DT_NAMES.get(run_dt, str(run_dt))
for name in names:
    fk = fetch.get(name, (0, 0.0))[1]
    wk = write.get(name, (0, 0.0))[1]
    entry = dict(kernel=name[:120], dispatches=fetch.get(name, write.get(name))[0], fetch_size_kib=fk, write_size_kib=wk,
                 hbm_gb_per_launch=(2 * fk + wk) * 1024 / 1e9)
    mm = re.search(r'gemm_kernel<(\d), d3r::GemmCfg<([0-9, ]+)>', name)
    p4 = 'gemm_p4_kernel<' in name        # the persistent split-fp16 kernel (csrc/gemm_p4.hip): tile configuration 10 of bench.py, one entry per epilogue instantiation
    if mm or p4:
        cfg = 10 if p4 else CFG_OF.get(tuple(int(x) for x in mm.group(2).split(',')))
        if cfg is not None and (p4 or int(mm.group(1)) == run_dt):
            # several instantiations share a tile-configuration id of bench.py (cfg 0 = the 128x128 tile by four waves AND by eight):
            # dispatch-weighted mean under the id, every instantiation listed
            prev = digest['gemm_cfg'].get(str(cfg))
            if prev is None:
                digest['gemm_cfg'][str(cfg)] = dict(entry, instantiations=[dict(entry)])
            else:
                n0, n1 = prev['dispatches'], entry['dispatches']
                for key in ('fetch_size_kib', 'write_size_kib', 'hbm_gb_per_launch'):
                    prev[key] = (prev[key] * n0 + entry[key] * n1) / max(n0 + n1, 1)
                prev['dispatches'] = n0 + n1
                prev['kernel'] = 'dispatch-weighted mean of %d instantiations' % (len(prev['instantiations']) + 1)
                prev['instantiations'].append(dict(entry))
    elif 'aligner_main_kernel' in name:
        digest['aligner_main_kernel'] = entry
    elif 'attention_kernel<' in name or 'attention_x3_kernel<' in name:
        digest['attention_kernel'] = entry
digest['visit'] = os.environ.get('D3R_VISIT', 'unnamed visit') + ': rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py --single-stream'
if fetch or write:
    with open(os.path.join(out, 'pmc_latest.json'), 'w') as f:
        json.dump(digest, f, indent=1)
