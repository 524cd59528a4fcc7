#!/bin/bash
# Round 4, visit L: per-shape launch tables of the forward with and without tile configuration 9 on ONE box (bench.py's profiled forward), twice each.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
for rep in 1 2; do for t in 0 2; do D3R_GEMM_T384=$t timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-fast --no-aligner --no-parity > $OUT/bench_t384_${t}_$rep.json 2> $OUT/bench_t384_${t}_$rep.log; grep "pairs/s on" $OUT/bench_t384_${t}_$rep.log; done; done
python - <<'PY'
import json
def tab(f):
    d = json.load(open(f)); return {(r['kernel'].split()[0], r['M'], r['N'], r['K'], r['kernel']): r for r in d['launch_table']}, d['value']
for rep in (1, 2):
    a, va = tab(f'gpurun_out/bench_t384_0_{rep}.json'); b, vb = tab(f'gpurun_out/bench_t384_2_{rep}.json')
    print(f'rep {rep}: {va:.1f} -> {vb:.1f} pairs/s')
    shapes = sorted({k[1:4] for k in list(a) + list(b) if k[1] == 24576})
    for s in shapes:
        ra = [r for k, r in a.items() if k[1:4] == s]; rb = [r for k, r in b.items() if k[1:4] == s]
        print('  ', s, ' | '.join(f"{r['kernel']} x{r['launches']} {r['ms']:.2f} ms" for r in ra), ' ==> ', ' | '.join(f"{r['kernel']} x{r['launches']} {r['ms']:.2f} ms" for r in rb))
PY
