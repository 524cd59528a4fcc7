"""Worker of the two-ranks-on-one-GPU test (tests/test_forward_gpu.py): one process per rank, BOTH on cuda:0, the real engine, gloo with CUDA
tensors as the transport (RCCL refuses two ranks on one device; its code path is covered at world size 1 on the same engine)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def scene_pairs(kind):
    import torch
    from dust3r_amd.image_pairs import make_pairs
    from dust3r_amd.synthetic import synthetic_image_list
    if kind == 'same':       # 5 views, complete symmetrised graph: 20 pairs sharing 5 images (the encode-once route)
        return make_pairs(synthetic_image_list(5, 32, 48, seed=8), 'complete', None, symmetrize=True)
    g = torch.Generator().manual_seed(4)
    shapes = [(32, 48), (48, 32), (32, 48), (32, 32), (48, 32)]
    imgs = [dict(img=torch.rand((1, 3, h, w), generator=g) * 2 - 1, true_shape=torch.tensor([[h, w]], dtype=torch.int32), idx=k, instance=str(k))
            for k, (h, w) in enumerate(shapes)]
    return [(imgs[i], imgs[j]) for i in range(len(imgs)) for j in range(len(imgs)) if i != j]


def build_engine(gpu):
    from dust3r_amd.model import AsymmetricCroCo3DStereo
    from dust3r_amd.synthetic import MODEL_CONFIGS
    from oracle.dust3r_ref import build_ref_model
    m = AsymmetricCroCo3DStereo(precision='fp16x3', landscape_only=False, **MODEL_CONFIGS['tiny_dpt'])
    m.load_state_dict(build_ref_model('tiny_dpt').state_dict(), strict=True)
    return m.to(gpu)


def worker(rank, world, port, outdir):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    gpu = torch.device('cuda', 0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from dust3r_amd.parallel import inference_sharded
        eng = build_engine(gpu)
        out = {}
        for kind in ('same', 'mixed'):
            out[kind] = inference_sharded(scene_pairs(kind), eng, gpu, batch_size=4)
        torch.save(out, os.path.join(outdir, f'rank{rank}.pt'))
        dist.barrier()
    finally:
        dist.destroy_process_group()


# ---------------------------------------------------------------------------------------------------- the alignment loop over two ranks on one GPU
ALIGN_SCENE = dict(n_views=7, H=96, W=128, seed=3, scene_graph='complete', symmetrize=True, noise=0.005)
ALIGN_NITER = 40


def aligned_scene(gpu, group, init, seed=11):
    import torch
    from dust3r_amd.cloud_opt import GlobalAlignerMode, global_aligner
    from dust3r_amd.synthetic import synthetic_scene
    out, state, _ = synthetic_scene(device=gpu, **ALIGN_SCENE)
    torch.manual_seed(seed)                                  # the random start of init=None (every process draws the same; rank 0's is broadcast anyway)
    scene = global_aligner(out, device=gpu, mode=GlobalAlignerMode.PointCloudOptimizer, verbose=False)
    if init == 'frozen':
        # a DIFFERENT random start per rank (seed) with the focals frozen and not preset: the frozen tensor must come from rank 0 like the trainable ones,
        # or the replicated pose step diverges silently between ranks (round-5 advisor item)
        scene.im_focals.requires_grad_(False)
        loss = scene.compute_global_alignment(init=None, niter=12, schedule='linear', lr=0.01, group=group)
        res = {k: getattr(scene, k).detach().cpu().clone() for k in ('pw_poses', 'pw_adaptors', 'im_poses', 'im_depthmaps', 'im_focals', 'im_pp')}
        res['loss'] = float(loss)
        return res
    if init == 'state':
        scene.load_state_dict(state)
        loss = scene.compute_global_alignment(init=None, niter=ALIGN_NITER, schedule='cosine', lr=0.01, group=group)
    else:
        loss = scene.compute_global_alignment(init='mst', niter=ALIGN_NITER, schedule='cosine', lr=0.01, group=group)
    res = {k: getattr(scene, k).detach().cpu().clone() for k in ('pw_poses', 'im_poses', 'im_depthmaps', 'im_focals')}
    res['loss'] = float(loss)
    return res


def align_worker(rank, world, port, outdir):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    gpu = torch.device('cuda', 0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        out = {init: aligned_scene(gpu, True, init) for init in ('state', 'mst')}
        out['frozen'] = aligned_scene(gpu, True, 'frozen', seed=100 + rank)
        torch.save(out, os.path.join(outdir, f'align_rank{rank}.pt'))
        dist.barrier()
    finally:
        dist.destroy_process_group()
