"""What the two-call form of the alignment loop costs per iteration on ONE rank (compute_global_alignment(group=...) at world size 1: step_begin -> all_reduce -> step_end
driven from Python) against d3r_aligner_run (all iterations enqueued by one C call), 100 views / 600 edges and 20 views / 190 edges. The difference is host-side enqueue time
per iteration -- the floor of the rank-shared loop whatever the number of ranks. Usage: python tools/sharded_loop_overhead.py"""
import os
import sys
import time

import torch

sys.path.insert(0, '.')


def main():
    import torch.distributed as dist
    dev = torch.device('cuda:0')
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29533')
    dist.init_process_group('nccl', rank=0, world_size=1)
    from dust3r_amd.cloud_opt import GlobalAlignerMode, global_aligner
    from dust3r_amd.synthetic import synthetic_scene
    for n, graph, sym in ((20, 'complete', False), (100, 'swin-3', True)):
        out, state, _ = synthetic_scene(n, 384, 512, seed=0, scene_graph=graph, symmetrize=sym, noise=0.002, device=dev, device_rng=True)
        res = {}
        for label, group in (('one C call (d3r_aligner_run)', None), ('two calls + all_reduce per iteration', True)):
            ts = []
            for rep in range(3):
                scene = global_aligner(out, device=dev, mode=GlobalAlignerMode.PointCloudOptimizer, verbose=False)
                scene.load_state_dict(state)
                torch.cuda.synchronize()
                t = time.perf_counter()
                loss = scene.compute_global_alignment(init=None, niter=300, schedule='cosine', lr=0.01, group=group)
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t)
                poses = scene.get_im_poses().detach().clone()
                del scene
            res[label] = (min(ts), loss, poses)
            print(f'  {n} views / {len(out["view1"]["idx"])} edges, {label:40s}: 300 iterations {min(ts) * 1e3:7.1f} ms = {min(ts) / 300 * 1e6:6.1f} us per iteration, loss {loss:.6f}', flush=True)
        (a, la, pa), (b, lb, pb) = res.values()
        print(f'  -> +{(b - a) / 300 * 1e6:.1f} us per iteration; same loss {la == lb}, same poses {bool(torch.equal(pa, pb))}')
        del out
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
