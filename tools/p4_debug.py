"""Debug aid for the persistent GEMM: where do its outputs differ from the one-tile-per-block kernels?"""
import math, os, sys
import torch
sys.path.insert(0, '.')
from dust3r_amd import ops
from dust3r_amd._lib import lib, ptr, current_stream, check, DTYPE_F16X3
dev = torch.device('cuda:0')
M, N, K, epi = [int(x) for x in (sys.argv[1:5] if len(sys.argv) > 4 else (4096, 256, 1024, 0))]
torch.manual_seed(0)
a = ops.pack_x3(torch.randn((M, K), device=dev))
w = ops.pad_rows(ops.pack_x3(torch.randn((N, K), device=dev) / math.sqrt(K)))
b = ops.pad_rows(torch.randn(N, device=dev))
res = ops.pack_x3(torch.randn((M, N), device=dev)) if epi == 3 else None
out = torch.zeros((M, 2 * N), dtype=torch.float16, device=dev)
part = torch.zeros((M, N // 32, 2), dtype=torch.float32, device=dev) if epi == 3 else None
def run():
    if epi == 3:
        check(lib.d3r_linear_x3res(ptr(a), ptr(w), ptr(b), ptr(out), ptr(res), ptr(part), M, N, K, current_stream()))
    else:
        check(lib.d3r_linear(ptr(a), ptr(w), ptr(b), ptr(out), None, M, N, K, epi, DTYPE_F16X3, current_stream()))
    torch.cuda.synchronize()
os.environ['D3R_GEMM_PERSIST'] = '0'; run(); ref = ops.unpack_x3(out.clone()) if hasattr(ops, 'unpack_x3') else out.clone().float(); refp = part.clone() if part is not None else None
out.zero_()
os.environ['D3R_GEMM_PERSIST'] = '1'; run(); got = ops.unpack_x3(out.clone()) if hasattr(ops, 'unpack_x3') else out.clone().float()
bad = (got != ref)
print(f'M={M} N={N} K={K} epi={epi} grid={os.environ.get("D3R_P4_GRID")}: mismatching elements {int(bad.sum())} of {bad.numel()}')
if bad.any():
    rows = bad.any(dim=1).nonzero().flatten()
    cols = bad.any(dim=0).nonzero().flatten()
    print('  bad rows:', rows[:8].tolist(), '...', rows[-4:].tolist(), 'count', len(rows))
    print('  bad cols:', cols[:8].tolist(), '...', cols[-4:].tolist(), 'count', len(cols))
    tm, tn = M // 256, got.shape[1] // 128
    g = bad.view(tm, 256, tn, -1).float().mean(dim=(1, 3))
    print('  fraction bad per tile (rows = tile m, cols = tile n):')
    for r in range(min(tm, 16)):
        print('   ', ' '.join(f'{float(x):.2f}' for x in g[r]))
    i = bad.nonzero()[0]
    print('  first bad', i.tolist(), float(got[i[0], i[1]]), float(ref[i[0], i[1]]))
if part is not None:
    print('  sums differ:', int((part != refp).sum()), 'of', part.numel())
