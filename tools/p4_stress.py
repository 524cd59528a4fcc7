"""On-GPU stress (not a test of the suite: minutes of GPU time): the persistent split-fp16 GEMM (csrc/gemm_p4.hip) against the one-tile-per-block kernels on random eligible
shapes and every epilogue it implements, each launch REPEATED under a concurrent load on a second stream (timing of DMA landings, barriers and store drains perturbed) --
every repetition must be BIT-equal to the reference launch. The kernel counts its own vmcnt and orders LDS reuse by hand; a miscounted wait shows up as a rare wrong
tile, which a single parity launch per shape can miss. Usage: python tools/p4_stress.py [shapes=24] [reps=12] [seed=0]"""
import math
import os
import sys

import torch

sys.path.insert(0, '.')
from dust3r_amd import ops  # noqa: E402
from dust3r_amd._lib import lib, ptr, current_stream, check, DTYPE_F16X3  # noqa: E402

dev = torch.device('cuda:0')


def launch(kind, a, w, b, res, out, part, M, N, K):
    if kind in ('store', 'gelu'):
        check(lib.d3r_linear(ptr(a), ptr(w), ptr(b), ptr(out), None, M, N, K, 0 if kind == 'store' else 2, DTYPE_F16X3, current_stream()))
    else:
        check(lib.d3r_linear_x3res(ptr(a), ptr(w), ptr(b), ptr(out), ptr(res) if 'res' in kind else None, ptr(part) if 'sums' in kind else None, M, N, K, current_stream()))


def main():
    n_shapes = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    g = torch.Generator().manual_seed(int(sys.argv[3]) if len(sys.argv) > 3 else 0)
    side = torch.cuda.Stream()
    la, lw = torch.randn((8192, 2048), device=dev), torch.randn((2048, 2048), device=dev)
    bad = 0
    for it in range(n_shapes):
        M = 256 * int(torch.randint(8, 129, (1,), generator=g))
        N = 128 * int(torch.randint(1, 25, (1,), generator=g))
        K = 32 * int(torch.randint(18, 129, (1,), generator=g))
        kind = ('store', 'gelu', 'x3', 'x3+res', 'x3+res+sums', 'x3+sums')[it % 6]
        a = ops.pack_x3(torch.randn((M, K), generator=g).to(dev))
        w = ops.pad_rows(ops.pack_x3((torch.randn((N, K), generator=g) / math.sqrt(K)).to(dev)))
        b = ops.pad_rows(torch.randn(N, generator=g).to(dev))
        res = ops.pack_x3(torch.randn((M, N), generator=g).to(dev))
        out = torch.empty((M, 2 * N), dtype=torch.float16, device=dev)
        part = torch.zeros((M, N // 32, 2), dtype=torch.float32, device=dev)
        os.environ['D3R_GEMM_PERSIST'] = '0'
        launch(kind, a, w, b, res, out, part, M, N, K)
        torch.cuda.synchronize()
        ref_o, ref_p = out.view(torch.int16).clone(), part.clone()
        os.environ['D3R_GEMM_PERSIST'] = '1'
        wrong = 0
        for r in range(reps):
            out.zero_()
            part.zero_()
            torch.cuda.synchronize()
            if r % 3:                       # two of three repetitions beside a load on another stream (a torch GEMM, an elementwise pass: other CUs' L2 / HBM traffic and clocks)
                with torch.cuda.stream(side):
                    for _ in range(1 + r % 4):
                        (la @ lw).relu_()
            launch(kind, a, w, b, res, out, part, M, N, K)
            torch.cuda.synchronize()
            ok = torch.equal(out.view(torch.int16), ref_o) and ('sums' not in kind or torch.equal(part, ref_p))
            wrong += 0 if ok else 1
        bad += wrong
        print(f'[{it:3d}] M={M:6d} N={N:5d} K={K:5d} {kind:12s} tiles={M // 256 * (N // 128):5d}: {reps - wrong}/{reps} bit-equal', flush=True)
        del a, w, res, out, part
    os.environ.pop('D3R_GEMM_PERSIST', None)
    print(f'p4 stress: {bad} mismatching launches of {n_shapes * reps}')
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()
