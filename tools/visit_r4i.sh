#!/bin/bash
# Round 4, visit I: the one-round 384 x 192 tile for the decoder's 24576-row GEMMs (D3R_GEMM_T384=1): bit-identity with the default tiles, per-shape timing, A/B on the forward.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
python - <<'PY' > $OUT/t384.log 2>&1
import os, time, torch, math
from dust3r_amd import ops
g = torch.Generator().manual_seed(1)
dev = torch.device('cuda:0')
def timeit(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / reps
for (M, N, K) in ((24576, 768, 768), (24576, 768, 3072), (24576, 3072, 768), (24576, 1536, 768), (24576, 2304, 768)):
    a = torch.randn((M, K), generator=g).to(dev); w = (torch.randn((N, K), generator=g) / math.sqrt(K)).to(dev); b = torch.randn(N, generator=g).to(dev); r = torch.randn((M, N), generator=g).to(dev)
    from dust3r_amd.ops import pack_x3, pad_rows, ptr, check, current_stream
    from dust3r_amd._lib import lib, DTYPE_F16X3
    ap, wp, bp = pack_x3(a), pad_rows(pack_x3(w)), pad_rows(b)
    outs, tm = [], []
    for t in ('0', '1'):
        os.environ['D3R_GEMM_T384'] = t
        res = []
        for epi, e in (('f32', 1), ('store', 0), ('gelu', 2)):
            out = torch.empty((M, N), dtype=torch.float32, device=dev) if e == 1 else torch.empty((M, 2 * N), dtype=torch.float16, device=dev)
            fn = lambda: check(lib.d3r_linear(ptr(ap), ptr(wp), ptr(bp), ptr(out), ptr(r) if e == 1 else None, M, N, K, e, DTYPE_F16X3, current_stream()), 'linear')
            ms = timeit(fn) * 1e3
            res.append((out.clone(), ms))
        outs.append(res)
    ok = all(torch.equal(x[0], y[0]) for x, y in zip(*outs))
    fl = 2.0 * M * N * K
    print(M, N, K, 'bit-identical' if ok else 'DIFFERENT', ' | '.join(f'{n}: {x[1] * 1e3:.1f} -> {y[1] * 1e3:.1f} us ({fl / x[1] / 1e9:.0f} -> {fl / y[1] / 1e9:.0f} TF/s)' for n, x, y in zip(('f32+res', 'store', 'gelu'), *outs)), flush=True)
PY
cat $OUT/t384.log | tail -8
for t in 0 1 0 1; do echo "D3R_GEMM_T384=$t"; D3R_GEMM_T384=$t timeout 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-fast --no-profile --no-aligner --no-parity 2>&1 | grep "pairs/s on"; done > $OUT/ab_t384.txt 2>&1; cat $OUT/ab_t384.txt
