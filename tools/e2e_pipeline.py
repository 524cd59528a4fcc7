"""End-to-end wall clock of the reference demo's pipeline (dust3r/demo.py:135-186 `get_reconstructed_scene`, without
the gradio UI / GLB export) on ONE MI355X -- BASELINE.json configs[4] at single-GPU scale:

    synthetic views -> make_pairs(swin-3, symmetrize) -> inference() -> global_aligner(PointCloudOptimizer)
    -> compute_global_alignment(init='mst', niter=300, schedule='cosine', lr=0.01) -> scene getters

Stage A (forward) runs on random images with random-init weights (what the throughput depends on); stage B (alignment) runs
on a geometrically consistent synthetic scene of the same size (random-weight pointmaps are not a scene), with the reference's
own initialisation (MST + Procrustes + PnP on the host, as in the reference) and the fused HIP optimisation loop.

Usage: python tools/e2e_pipeline.py [n_views=100] [scene_graph=swin-3] [H=384] [W=512]
"""
import sys
import time

import torch

sys.path.insert(0, '.')


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    graph = sys.argv[2] if len(sys.argv) > 2 else 'swin-3'
    H = int(sys.argv[3]) if len(sys.argv) > 3 else 384
    W = int(sys.argv[4]) if len(sys.argv) > 4 else 512
    dev = torch.device('cuda:0')
    from dust3r_amd.cloud_opt import GlobalAlignerMode, global_aligner
    from dust3r_amd.image_pairs import make_pairs
    from dust3r_amd.inference import inference
    from dust3r_amd.model import AsymmetricCroCo3DStereo
    from dust3r_amd.synthetic import MODEL_CONFIGS, OUT_GAIN, synthetic_image_list, synthetic_scene, synthetic_state_dict
    cfg = 'DUSt3R_ViTLarge_BaseDecoder_512_dpt'
    if '--align-only' in sys.argv:
        return align_stage(n, graph, H, W, dev)
    prec = next((a.split('=')[1] for a in sys.argv if a.startswith('--precision=')), None)      # default: the engine default (fp16x3, parity-grade)
    m = AsymmetricCroCo3DStereo(precision=prec, landscape_only=False, **MODEL_CONFIGS[cfg])
    m.load_state_dict(synthetic_state_dict({k: torch.empty(v, device='meta') for k, v in m._spec.items()}, 0, OUT_GAIN[cfg], device=dev))
    m.to(dev)
    print(f'== end to end: {n} views {H}x{W}, scene graph {graph} (symmetrised), {cfg}, {m.precision}')

    imgs = synthetic_image_list(n, H, W, seed=0)
    pairs = make_pairs(imgs, scene_graph=graph, prefilter=None, symmetrize=True)
    # ---- stage A: inference (host images in, host predictions out, like the reference) ---------------------------------
    for mode, kw in (('encode-once, host outputs (default)', dict(encode_once=True)), ('encode-once, outputs stay in HBM', dict(encode_once=True, output_device=dev)),
                     ('pair-by-pair, host outputs', dict(encode_once=False))):
        inference(pairs[:16], m, dev, batch_size=8, verbose=False, **kw)        # warm-up (workspace allocation)
        torch.cuda.synchronize()
        t = time.time()
        out = inference(pairs, m, dev, batch_size=32, verbose=False, **kw)
        torch.cuda.synchronize()
        dt = time.time() - t
        print(f'  inference, {mode:36s}: {len(pairs)} pairs in {dt:6.2f} s = {len(pairs) / dt:6.1f} pairs/s (host to host)')
    del out

    align_stage(n, graph, H, W, dev)


def align_stage(n, graph, H, W, dev):
    from dust3r_amd.cloud_opt import GlobalAlignerMode, global_aligner
    from dust3r_amd.synthetic import synthetic_scene
    # ---- stage B: global alignment of a consistent scene of the same shape ------------------------------------------------
    t = time.time()
    out, _, gt = synthetic_scene(n, H, W, seed=0, scene_graph=graph, symmetrize=True, noise=0.002, device=dev)
    host_out = '--host-predictions' in sys.argv     # predictions handed over on the host (the reference's inference() format) or resident in HBM
    if host_out:
        out = {k: ({kk: (vv.cpu() if isinstance(vv, torch.Tensor) else vv) for kk, vv in v.items()} if isinstance(v, dict) else v) for k, v in out.items()}
    print(f'  (synthetic scene built in {time.time() - t:.1f} s: {len(out["view1"]["idx"])} edges; predictions {"on the host" if host_out else "resident in HBM"})')
    reps = int(next((a.split('=')[1] for a in sys.argv if a.startswith('--repeat=')), 1))
    for rep in range(reps):
        _align_once(out, gt, dev, last=rep == reps - 1)


def _align_once(out, gt, dev, last):
    from dust3r_amd.cloud_opt import GlobalAlignerMode, global_aligner
    torch.cuda.synchronize()
    t0 = time.time()
    if '--profile-build' in sys.argv and last:
        import cProfile
        import pstats
        pr = cProfile.Profile()
        pr.enable()
    scene = global_aligner(out, device=dev, mode=GlobalAlignerMode.PointCloudOptimizer, verbose=False)
    torch.cuda.synchronize()
    t1 = time.time()
    if '--profile-build' in sys.argv and last:
        pr.disable()
        pstats.Stats(pr).sort_stats('cumulative').print_stats(22)
    from dust3r_amd.cloud_opt import init_im_poses as init_fun
    if '--profile-init' in sys.argv and last:
        import cProfile
        import pstats
        pr = cProfile.Profile()
        pr.enable()
    init_fun.init_minimum_spanning_tree(scene, niter_PnP=10)
    torch.cuda.synchronize()
    t2 = time.time()
    if '--profile-init' in sys.argv and last:
        pr.disable()
        pstats.Stats(pr).sort_stats('cumulative').print_stats(30)
    loss = scene.compute_global_alignment(init=None, niter=300, schedule='cosine', lr=0.01)
    torch.cuda.synchronize()
    t3 = time.time()
    poses, focals, pts = scene.get_im_poses(), scene.get_focals(), scene.get_pts3d()
    masks = scene.get_masks()
    torch.cuda.synchronize()
    t4 = time.time()
    print(f'  global_aligner(): build {t1 - t0:5.2f} s | init=mst (GPU bootstrap) {t2 - t1:6.2f} s | 300 iterations {t3 - t2:5.2f} s = {300 / (t3 - t2):6.1f} it/s | getters {t4 - t3:4.2f} s')
    print(f'  final loss {loss:.5f}; focal error vs ground truth {float((focals.flatten().cpu() / gt["focal"] - 1).abs().max()):.3f}; '
          f'{len(pts)} pointmaps of {tuple(pts[0].shape)}, {sum(int(mk.sum()) for mk in masks)} confident points, poses {tuple(poses.shape)}')


if __name__ == '__main__':
    main()
