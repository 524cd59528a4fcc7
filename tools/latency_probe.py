"""GPU probe (not product, not a test): where is the engine launch-bound rather than throughput-bound?
 * forward at B = 1, 2, 4, 8, 32 pairs per call (the reference demo calls inference() with batch_size = 1);
 * aligner iterations/s on small scenes (PairViewer-size to the BASELINE scene).
Usage (GPU box): python tools/latency_probe.py"""
import sys
import time

import torch

sys.path.insert(0, '.')
from bench import build_model, H, W  # noqa: E402
from dust3r_amd.synthetic import synthetic_scene, synthetic_views  # noqa: E402


def timed_forward(model, B, dev):
    v1, v2 = synthetic_views(B, H, W, seed=0, device=dev)
    for _ in range(4):
        model(v1, v2)
    torch.cuda.synchronize()
    n = 20 if B <= 8 else 5
    t = time.perf_counter()
    for _ in range(n):
        model(v1, v2)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n


def main():
    import os
    dev = torch.device('cuda', 0)
    model = build_model('fp16x3', dev)
    if 'small-tiles' in sys.argv:
        # the 64 x 64 GEMM tile of the small-batch forwards: crossover sweep (D3R_GEMM_T64 = 128x128-tile count below which it is taken; 0 = never)
        for ns in ('2', '3', '4'):
            os.environ['D3R_GEMM_64NS'] = ns
            for t64 in ('0', '200', '400', '1000'):
                os.environ['D3R_GEMM_T64'] = t64
                print(f'D3R_GEMM_64NS={ns} D3R_GEMM_T64={t64:5s}: ' + '  '.join(f'B={B}: {timed_forward(model, B, dev) * 1e3:7.2f} ms' for B in (1, 2, 3, 4, 6, 8)), flush=True)
        os.environ.pop('D3R_GEMM_T64')
        os.environ.pop('D3R_GEMM_64NS')
        # where the one-pair forward spends its time (event-profiled launches; adds event overhead)
        from bench import read_launch_table, read_profile
        for t64 in ('0', None):
            if t64 is not None:
                os.environ['D3R_GEMM_T64'] = t64
            else:
                os.environ.pop('D3R_GEMM_T64', None)
            v1, v2 = synthetic_views(1, H, W, seed=0, device=dev)
            from dust3r_amd._lib import lib
            lib.d3r_model_set_option(model._engine, 1, 1)
            model(v1, v2)
            torch.cuda.synchronize()
            prof, rows = read_profile(model), read_launch_table(model)
            lib.d3r_model_set_option(model._engine, 1, 0)
            print(f'-- one pair, D3R_GEMM_T64={t64}: linear {prof["linear"]}, conv {prof["conv"]}, attention {prof["attention"]}, other {prof["other"]}')
            for r in rows[:28]:
                print('   ', r)
        return
    for graphs in (0, 4):              # eager launches vs the hipGraph replay of small forwards (D3R_MODEL_OPT_GRAPH_MAX_PAIRS)
        model.set_graph_max_pairs(graphs)
        for B in ((1, 2, 4, 8, 32) if graphs == 0 else (1, 2, 4)):
            dt = timed_forward(model, B, dev)
            print(f"forward B={B:2d} {'graph replay' if graphs else 'eager       '}: {dt * 1e3:8.2f} ms/call  {B / dt:7.1f} pairs/s", flush=True)
    if 'forward-only' in sys.argv:
        return
    del model
    from dust3r_amd.cloud_opt import global_aligner
    from dust3r_amd.cloud_opt.base_opt import global_alignment_loop
    for (n, h, w) in ((2, 224, 224), (4, 224, 224), (4, 384, 512), (8, 384, 512), (20, 384, 512)):
        out, init, gt = synthetic_scene(n, h, w, seed=0, symmetrize=True)
        scene = global_aligner(out, dev, verbose=False)
        scene.load_state_dict(init)
        global_alignment_loop(scene, lr=0.01, niter=50, schedule='cosine', lr_min=1e-6)
        torch.cuda.synchronize()
        t = time.perf_counter()
        global_alignment_loop(scene, lr=0.01, niter=300, schedule='cosine', lr_min=1e-6)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        print(f'aligner n={n:2d} E={scene.n_edges:3d} {h}x{w}: {300 / dt:8.0f} iters/s  ({dt / 300 * 1e6:6.1f} us/iter)', flush=True)


if __name__ == '__main__':
    main()
