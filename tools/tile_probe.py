"""On-GPU probe (not a test): split-fp16 GEMMs of the 32-pair step, per tile configuration, with and without their epilogue.
What a tile shape's K loop delivers alone (D3R_GEMM_NOSTORE, probe builds only) against what the launch delivers with its epilogue is the
room an overlapped epilogue has on that shape. Usage: python tools/tile_probe.py [cfgs, e.g. a,1,2,7,p]   (a = heuristic, p = persistent)"""
import math
import os
import sys

import torch

sys.path.insert(0, '.')
from dust3r_amd import ops  # noqa: E402
from dust3r_amd._lib import lib, ptr, current_stream, check, DTYPE_F16X3  # noqa: E402

dev = torch.device('cuda:0')


def timeit(fn, warm=2, reps=8):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(reps):
        fn()
    ev1.record()
    torch.cuda.synchronize()
    return ev0.elapsed_time(ev1) / reps


# epilogue codes of d3r_linear (0 typed store, 1 fp32 + fp32 residual, 2 GELU); 3 = d3r_linear_x3res: typed residual stream + LayerNorm partial sums (the default engine's proj / fc2)
SHAPES = [(49152, 4096, 1024, 2, 'enc fc1 + GELU'), (49152, 3072, 1024, 0, 'enc qkv-sized plain store'), (49152, 1024, 1024, 3, 'enc proj + typed residual + sums'),
          (49152, 1024, 4096, 3, 'enc fc2 + typed residual + sums'), (24576, 3072, 768, 2, 'dec fc1 + GELU'), (24576, 768, 768, 3, 'dec proj + typed residual + sums'),
          (24576, 768, 3072, 3, 'dec fc2 + typed residual + sums'), (24576, 2304, 768, 0, 'dec qkv-sized plain store')]
if os.environ.get('D3R_PROBE_CUSTOM'):        # "M,N,K,epi;M,N,K,epi;..." instead of the 32-pair step's shapes (e.g. the one-pair call's: 1536,1024,4096,3)
    SHAPES = [tuple(int(x) for x in t.split(',')) + (f'custom {t}',) for t in os.environ['D3R_PROBE_CUSTOM'].split(';')]
if os.environ.get('D3R_PROBE_SHAPES'):
    SHAPES = [SHAPES[int(i)] for i in os.environ['D3R_PROBE_SHAPES'].split(',')]


def main():
    cfgs = (sys.argv[1].split(',') if len(sys.argv) > 1 else ['a', '1', '2', '7'])
    nostore = os.environ.get('D3R_PROBE_NOSTORE', '1') == '1'
    print('== split-fp16 GEMM: ms / TFLOP/s per tile configuration; "ns" = the same launch without its epilogue (probe builds)')
    for (M, N, K, epi, name) in SHAPES:
        a = ops.pack_x3(torch.randn((M, K), device=dev))
        w = ops.pad_rows(ops.pack_x3(torch.randn((N, K), device=dev) / math.sqrt(K)))
        b = ops.pad_rows(torch.randn(N, device=dev))
        res = torch.randn((M, N), device=dev) if epi == 1 else ops.pack_x3(torch.randn((M, N), device=dev)) if epi == 3 else None
        out = torch.empty((M, N), dtype=torch.float32, device=dev) if epi == 1 else torch.empty((M, 2 * N), dtype=torch.float16, device=dev)
        part = torch.zeros((M, N // 32, 2), dtype=torch.float32, device=dev) if epi == 3 else None

        def run():
            if epi == 3:
                check(lib.d3r_linear_x3res(ptr(a), ptr(w), ptr(b), ptr(out), ptr(res), ptr(part), M, N, K, current_stream()))
            else:
                check(lib.d3r_linear(ptr(a), ptr(w), ptr(b), ptr(out), ptr(res), M, N, K, epi, DTYPE_F16X3, current_stream()))
        line = f'  {name:28s} M={M} N={N} K={K}'
        ref = None
        for rnd in range(2):            # two interleaved rounds: the second is the one reported (clocks settled)
            cells = []
            for cfg in cfgs:
                os.environ.pop('D3R_GEMM_CFG', None)
                os.environ.pop('D3R_GEMM_PERSIST', None)
                if cfg == 'p':
                    os.environ['D3R_GEMM_PERSIST'] = '1'
                elif cfg == 'q':
                    os.environ['D3R_GEMM_PERSIST'] = '0'
                elif cfg != 'a':
                    os.environ['D3R_GEMM_CFG'] = cfg
                ms = timeit(run)
                cell = f'cfg {cfg}: {ms:6.3f} ms {2 * M * N * K / ms / 1e9:6.1f}'
                if cfg in ('p', 'q') and rnd == 1:
                    out.zero_()
                    if part is not None:
                        part.zero_()
                    run()
                    o = out.clone()
                    op = part.clone() if part is not None else None
                    if ref is None:
                        os.environ['D3R_GEMM_PERSIST'] = '0'
                        run()
                        ref = (out.clone(), part.clone() if part is not None else None)
                        os.environ['D3R_GEMM_PERSIST'] = '1' if cfg == 'p' else '0'
                    same = torch.equal(o.view(torch.int16), ref[0].view(torch.int16)) and (op is None or torch.equal(op, ref[1]))
                    if same:
                        cell += ' bit-equal'
                    else:
                        bad = (o.view(torch.int16) != ref[0].view(torch.int16))
                        cell += f' DIFF {int(bad.sum())} of {bad.numel()} halves, max {float((o.float() - ref[0].float()).abs().max()):.3e}'
                        if op is not None:
                            cell += f', sums {int((op != ref[1]).sum())} differ'
                if nostore:
                    os.environ['D3R_GEMM_NOSTORE'] = '1'
                    ms2 = timeit(run)
                    os.environ.pop('D3R_GEMM_NOSTORE')
                    cell += f' (ns {ms2:6.3f} {2 * M * N * K / ms2 / 1e9:6.1f})'
                cells.append(cell)
            os.environ.pop('D3R_GEMM_CFG', None)
            os.environ.pop('D3R_GEMM_PERSIST', None)
        print(line + ' | ' + ' | '.join(cells), flush=True)
        del a, w, out, res


if __name__ == '__main__':
    main()
