"""A/B of the folded LayerNorm (D3R_LN_FOLD, read at engine creation) on the driver-form forward: TWO engines with the same synthetic weights in one
process, alternating repetitions; forward ms (two streams), per-class ms of a profiled single-stream forward, the difference of the outputs, and the
one-pair latency of both. Usage: python tools/fold_probe.py [--reps=3] [--pairs=32]"""
import os
import sys
import time

sys.path.insert(0, '.')
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    reps = int(next((a.split('=')[1] for a in sys.argv if a.startswith('--reps=')), 3))
    B = int(next((a.split('=')[1] for a in sys.argv if a.startswith('--pairs=')), 32))
    dev = torch.device('cuda:0')
    from dust3r_amd.synthetic import synthetic_views
    eng = {}
    for f in ('0', '1'):
        os.environ['D3R_LN_FOLD'] = f
        eng[f] = bench.build_model('fp16x3', dev)
    v1, v2 = synthetic_views(B, bench.H, bench.W, seed=0, device=dev)
    s1, s2 = synthetic_views(1, bench.H, bench.W, seed=1, device=dev)

    def measure(m, a, b, n=5):
        for _ in range(2):
            m(a, b)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(n):
            m(a, b)
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / n * 1e3

    acc = {'0': [], '1': []}
    tables = {}
    for r in range(reps):
        for f in ('0', '1'):
            m = eng[f]
            ms = measure(m, v1, v2)
            m.set_two_streams(False)
            blk = bench.profile_mode(m, v1, v2, 'fp16x3', quiet=True)
            m.set_two_streams(True)
            k = blk['kernels']
            tables.setdefault(f, []).append({(t['kernel'], t['M'], t['N'], t['K']): (t['launches'], t['ms']) for t in blk['launch_table']})
            one = measure(m, s1, s2, n=20)
            row = (ms, k['attention']['ms'], k['other']['ms'], k['other']['launches'], k['all_gemm_linear']['ms'], k['all_gemm_conv']['ms'], one)
            acc[f].append(row)
            print(f'fold={f} rep {r}: forward {row[0]:8.2f} ms ({B / row[0] * 1e3:6.1f} pairs/s) | attention {row[1]:6.2f} | other {row[2]:6.2f} ({row[3]} launches) | linear {row[4]:7.2f} | conv {row[5]:6.2f} | one pair {row[6]:6.2f} ms', flush=True)
    for f in ('0', '1'):
        a = acc[f]
        mean = [sum(x[i] for x in a) / len(a) for i in range(7)]
        print(f'fold={f} MEAN : forward {mean[0]:8.2f} ms ({B / mean[0] * 1e3:6.1f} pairs/s) | attention {mean[1]:6.2f} | other {mean[2]:6.2f} ({mean[3]:.0f} launches) | linear {mean[4]:7.2f} | conv {mean[5]:6.2f} | one pair {mean[6]:6.2f} ms')
    o0 = eng['0'](v1, v2)
    o1 = eng['1'](v1, v2)
    a = torch.cat((o0[0]['pts3d'], o0[1]['pts3d_in_other_view']))
    b = torch.cat((o1[0]['pts3d'], o1[1]['pts3d_in_other_view']))
    rel = ((a - b).norm(dim=-1) / a.norm(dim=-1).clamp_min(1e-8)).flatten()
    print(f'folded vs LayerNorm kernels, {B} pairs: per-pixel rel diff max {float(rel.max()):.3e} p99.99 {float(rel.kthvalue(int(0.9999 * rel.numel())).values):.3e} mean {float(rel.mean()):.3e}')
    keys = sorted(set(tables['0'][0]) | set(tables['1'][0]), key=lambda kk: -tables['0'][0].get(kk, (0, 0.0))[1])
    print('per shape, ms per step (mean over repetitions): LayerNorm kernels -> folded')
    for kk in keys:
        a0 = [t[kk][1] for t in tables['0'] if kk in t]
        a1 = [t[kk][1] for t in tables['1'] if kk in t]
        if a0 and a1:
            print(f'    {kk[0]:12s} M={kk[1]:8d} N={kk[2]:5d} K={kk[3]:5d} x{tables["0"][0][kk][0]:3d}: {sum(a0) / len(a0):7.3f} -> {sum(a1) / len(a1):7.3f} ms')


if __name__ == '__main__':
    main()
