#!/bin/bash
# Round 4, visit F: split-fp16 256x256 tile with the DMA pieces interleaved with the MFMA rows (D3R_GEMM_X3IL=1): bit-identity with the plain loop, A/B on the forward.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
python - <<'PY' > $OUT/x3il_bitident.log 2>&1
import os, torch, math
from dust3r_amd import ops
g = torch.Generator().manual_seed(1)
dev = torch.device('cuda:0')
os.environ['D3R_GEMM_CFG'] = '1'
for (M, N, K) in ((2048, 1024, 1024), (515, 768, 768), (4096, 3072, 1024), (1000, 1024, 4096)):
    a = torch.randn((M, K), generator=g).to(dev); w = (torch.randn((N, K), generator=g) / math.sqrt(K)).to(dev); b = torch.randn(N, generator=g).to(dev); r = torch.randn((M, N), generator=g).to(dev)
    outs = []
    for il in ('0', '1'):
        os.environ['D3R_GEMM_X3IL'] = il
        outs.append([ops.linear_x3(a, w, b, e, residual=(r if e == 'f32' else None)).clone() for e in ('f32', 'store', 'gelu')])
    ok = all(torch.equal(x, y) for x, y in zip(*outs))
    ref = (a.double() @ w.double().T + b.double() + r.double()).float()
    print(M, N, K, 'bit-identical' if ok else 'DIFFERENT', 'rel err vs fp64', float((outs[1][0] - ref).abs().max() / ref.abs().max()))
PY
cat $OUT/x3il_bitident.log | tail -6
for il in 0 1 0 1; do echo "D3R_GEMM_X3IL=$il"; D3R_GEMM_X3IL=$il timeout 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-fast --no-profile --no-aligner --no-parity 2>&1 | grep "pairs/s on"; done > $OUT/ab_x3il.txt 2>&1; cat $OUT/ab_x3il.txt
