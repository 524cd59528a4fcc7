"""Probe: persistent GEMM, typed store on the qkv-sized shape, with parts of the drain removed (D3R_P4_DBG; results invalid)."""
import math, os, sys
import torch
sys.path.insert(0, '.')
from dust3r_amd import ops
from dust3r_amd._lib import lib, ptr, current_stream, check, DTYPE_F16X3
dev = torch.device('cuda:0')
M, N, K = 49152, 3072, 1024
a = ops.pack_x3(torch.randn((M, K), device=dev)); w = ops.pad_rows(ops.pack_x3(torch.randn((N, K), device=dev) / math.sqrt(K))); b = ops.pad_rows(torch.randn(N, device=dev))
out = torch.empty((M, 2 * N), dtype=torch.float16, device=dev)
def run():
    check(lib.d3r_linear(ptr(a), ptr(w), ptr(b), ptr(out), None, M, N, K, 0, DTYPE_F16X3, current_stream()))
def timeit(reps=10):
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
os.environ['D3R_GEMM_PERSIST'] = '1'
for rnd in range(2):
    line = []
    for tag, env in (('full', {}), ('no stores', {'D3R_P4_DBG': '1'}), ('no swaps', {'D3R_P4_DBG': '2'}), ('store early', {'D3R_P4_DBG': '4'}), ('plain stores', {'D3R_P4_DBG': '5'}), ('no DMA wait', {'D3R_P4_DBG': '6'}), ('NF=2', {'D3R_P4_NF': '2'}), ('K loop only', {'D3R_GEMM_NOSTORE': '1'}), ('cfg1', {'D3R_GEMM_PERSIST': '0'})):
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        ms = timeit()
        for k, v in old.items():
            if v is None: os.environ.pop(k)
            else: os.environ[k] = v
        line.append(f'{tag}: {ms:.3f} ms {2 * M * N * K / ms / 1e9:.0f}')
    print(' | '.join(line), flush=True)
