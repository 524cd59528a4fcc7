"""One pair per call through the PUBLIC API, host to host (dust3r/demo.py:156 with two images, visloc.py:88): `inference([pair], model, device, batch_size=1)`
returns CPU tensors; then `global_aligner(mode=PairViewer)`. Against the engine call on resident tensors (bench.py's latency block).
Usage: python tools/one_pair_host_probe.py [--profile]"""
import sys
import time

import torch

sys.path.insert(0, '.')


def main():
    dev = torch.device('cuda:0')
    import bench
    from dust3r_amd.cloud_opt import GlobalAlignerMode, global_aligner
    from dust3r_amd.image_pairs import make_pairs
    from dust3r_amd.inference import inference
    from dust3r_amd.synthetic import synthetic_image_list, synthetic_views
    model = bench.build_model('fp16x3', dev)
    imgs = synthetic_image_list(2, bench.H, bench.W, seed=0)
    v1, v2 = synthetic_views(1, bench.H, bench.W, seed=0, device=dev)

    def timed(label, fn, reps=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(reps):
            r = fn()
        torch.cuda.synchronize()
        print(f'  {label:72s} {(time.perf_counter() - t) / reps * 1e3:8.2f} ms', flush=True)
        return r

    timed('engine forward, one pair, tensors resident in HBM', lambda: model(v1, v2))
    one = [(imgs[0], imgs[1])]
    timed('inference([pair]) host to host (1 pair, batch_size=1)', lambda: inference(one, model, dev, batch_size=1, verbose=False))
    timed('inference([pair], output_device=cuda)', lambda: inference(one, model, dev, batch_size=1, verbose=False, output_device=dev))
    sym = make_pairs(imgs, scene_graph='complete', prefilter=None, symmetrize=True)
    out = timed('inference(2 symmetrised pairs) host to host (the demo with two images)', lambda: inference(sym, model, dev, batch_size=1, verbose=False))
    timed('global_aligner(mode=PairViewer) on that output', lambda: global_aligner(out, device=dev, mode=GlobalAlignerMode.PairViewer, verbose=False), reps=10)
    if '--profile' in sys.argv:
        import cProfile
        import pstats
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(20):
            inference(one, model, dev, batch_size=1, verbose=False)
        pr.disable()
        pstats.Stats(pr).sort_stats('cumulative').print_stats(35)


if __name__ == '__main__':
    main()
