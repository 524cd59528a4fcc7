"""Same-process A/B of switches the engine reads AT CREATION (D3R_DEC_KV_AHEAD, D3R_LN_INLINE_ROWS, D3R_ENC_SPLIT, D3R_GRAPH_MAX_PAIRS ...)
on the small-batch forwards of the bench's latency block: one engine per arm on the same weights, alternating repetitions, back-to-back calls of
1 / 2 / 4 / 8 (and optionally more) pairs of 512x384 in the default precision; the outputs of the arms are compared bit for bit.
Usage: python tools/latency_probe.py VAR=a,b[,c] [--reps=3] [--pairs=1,2,4,8]"""
import os
import sys
import time

sys.path.insert(0, '.')
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    spec = next(a for a in sys.argv[1:] if '=' in a and not a.startswith('--'))
    var, vals = spec.split('=')
    vals = vals.split(',')
    reps = int(next((a.split('=')[1] for a in sys.argv if a.startswith('--reps=')), 3))
    sizes = [int(x) for x in next((a.split('=')[1] for a in sys.argv if a.startswith('--pairs=')), '1,2,4,8').split(',')]
    dev = torch.device('cuda:0')
    from dust3r_amd.synthetic import synthetic_views
    engines = {}
    for v in vals:
        os.environ[var] = v
        engines[v] = bench.build_model('fp16x3', dev)
    os.environ.pop(var, None)
    v1, v2 = synthetic_views(max(sizes), bench.H, bench.W, seed=0, device=dev)
    sub = lambda d, n: {k: x[:n] for k, x in d.items()}  # noqa: E731
    acc = {(v, n): [] for v in vals for n in sizes}
    for r in range(reps):
        for n in sizes:
            a, b = sub(v1, n), sub(v2, n)
            calls = max(4, 24 // n)
            for v in vals:
                m = engines[v]
                for _ in range(3):
                    m(a, b)
                torch.cuda.synchronize()
                t = time.perf_counter()
                for _ in range(calls):
                    m(a, b)
                torch.cuda.synchronize()
                acc[(v, n)].append((time.perf_counter() - t) / calls * 1e3)
    print(f'== {var}: arms {vals}, {reps} alternating repetitions, ms per call (min / mean)')
    for n in sizes:
        print(f'   {n:3d} pairs: ' + ' | '.join(f'{var}={v}: {min(acc[(v, n)]):7.3f} / {sum(acc[(v, n)]) / reps:7.3f}' for v in vals))
    for n in sizes:
        a, b = sub(v1, n), sub(v2, n)
        outs = []
        for v in vals:
            o1, o2 = engines[v](a, b)
            outs.append((o1['pts3d'].clone(), o1['conf'].clone(), o2['pts3d_in_other_view'].clone(), o2['conf'].clone()))
        same = all(all(torch.equal(x, y) for x, y in zip(outs[0], o)) for o in outs[1:])
        print(f'   {n:3d} pairs: outputs of all arms ' + ('bit-identical' if same else 'DIFFER'))


if __name__ == '__main__':
    main()
