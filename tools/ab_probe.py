"""In-process A/B of per-launch environment switches on the driver-form forward (32 pairs 512x384, default precision): ONE engine, the
switch flipped between forwards, alternating repetitions -- a same-box, same-process comparison that costs seconds instead of one bench.py
run per arm. Only for switches the library reads on every launch (D3R_LN_PAIR, D3R_UPSAMPLE_XCD, D3R_ATTN_SC, D3R_ATTN_DMA, D3R_HEAD_FUSE ...).
Per arm: forward ms (two-stream schedule, 5 timed forwards after 2) and the per-class milliseconds of one profiled single-stream forward.
Usage: python tools/ab_probe.py VAR=a,b [VAR2=a,b ...] [--reps 3] [--pairs 32]"""
import os
import sys
import time

sys.path.insert(0, '.')
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    args = [a for a in sys.argv[1:] if '=' in a and not a.startswith('--')]
    reps = int(next((a.split('=')[1] for a in sys.argv if a.startswith('--reps=')), 3))
    B = int(next((a.split('=')[1] for a in sys.argv if a.startswith('--pairs=')), 32))
    dev = torch.device('cuda:0')
    from dust3r_amd.synthetic import synthetic_views
    model = bench.build_model('fp16x3', dev)
    v1, v2 = synthetic_views(B, bench.H, bench.W, seed=0, device=dev)

    def measure():
        for _ in range(2):
            model(v1, v2)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(5):
            model(v1, v2)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t) / 5 * 1e3
        model.set_two_streams(False)
        blk = bench.profile_mode(model, v1, v2, 'fp16x3', quiet=True)
        model.set_two_streams(True)
        k = blk['kernels']
        return ms, k['attention']['ms'], k['other']['ms'], k['all_gemm_linear']['ms'], k['all_gemm_conv']['ms']

    ref_out = None
    for spec in args:
        var, vals = spec.split('=')
        vals = vals.split(',')
        print(f'== {var}: arms {vals}, {reps} alternating repetitions, {B} pairs per forward')
        acc = {v: [] for v in vals}
        outs = {}
        for r in range(reps):
            for v in vals:
                os.environ[var] = v
                m = measure()
                acc[v].append(m)
                if r == 0:
                    o1, o2 = model(v1, v2)
                    outs[v] = (o1['pts3d'][:2].clone(), o2['conf'][:2].clone())
                print(f'   {var}={v} rep {r}: forward {m[0]:8.2f} ms ({B / m[0] * 1e3:6.1f} pairs/s) | attention {m[1]:6.2f} | other {m[2]:6.2f} | linear {m[3]:7.2f} | conv {m[4]:6.2f}', flush=True)
        os.environ.pop(var, None)
        for v in vals:
            a = acc[v]
            mean = [sum(x[i] for x in a) / len(a) for i in range(5)]
            print(f'   {var}={v} MEAN : forward {mean[0]:8.2f} ms ({B / mean[0] * 1e3:6.1f} pairs/s) | attention {mean[1]:6.2f} | other {mean[2]:6.2f} | linear {mean[3]:7.2f} | conv {mean[4]:6.2f}')
        base = outs[vals[0]]
        for v in vals[1:]:
            same = torch.equal(base[0], outs[v][0]) and torch.equal(base[1], outs[v][1])
            d = float((base[0] - outs[v][0]).abs().max())
            print(f'   outputs {var}={v} vs {var}={vals[0]}: {"bit-identical" if same else f"max abs diff {d:.3e}"}')


if __name__ == '__main__':
    main()
