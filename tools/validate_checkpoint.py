"""One command to validate the engine against the CPU oracle on a REAL checkpoint, the moment one is available (none is reachable from the
build container: no network, no .pth on disk -- the croco / roma restatements in oracle/ stay "parity unpinned" until this has run once).

    python tools/validate_checkpoint.py <DUSt3R_ViTLarge_BaseDecoder_512_dpt.pth | hub snapshot directory> [--size 512x384] [--pairs 2] [--images a.png b.png ...]
                                        [--precision fp16x3,fp32] [--align]

What it does (GPU box; test infrastructure like everything that imports oracle/):
  1. loads the checkpoint with the reference's own protocol (dust3r/model.py:27-43: ckpt['args'].model string, strict=False, ManyAR patch
     embed swapped) into BOTH the CPU oracle (oracle/dust3r_ref.py, the restated croco modules) and the engine (dust3r_amd.model.load_model);
     reports missing / unexpected keys of both loads -- a key-for-key, shape-for-shape match of the restated module tree with the released
     state dict is the first pin of the croco restatement;
  2. runs the same image pairs (files through dust3r_amd.utils.image.load_images, or seeded synthetic images) through both and prints
     SURVEY.md 8(d)'s statistics per engine precision: per-pixel ||pts_hip - pts_ref|| / max(||pts_ref||, 1e-8) max / p99.99 / p99 / mean for
     both views, relative confidence error, and min |pts| / mean |pts| (how close the pointmaps come to the origin);
  3. with --align: make_pairs(complete) over the images -> inference() -> global_aligner + 300 iterations on the engine, and the oracle's
     aligner (oracle/aligner_ref.py) from the same initial state for 30 iterations: loss trajectories side by side.
Exit code 0 when every requested precision's per-pixel max is <= 1e-3 (the north-star bar), 1 otherwise.
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def stats(got, ref):
    e = ((got.float().cpu() - ref).norm(dim=-1) / ref.norm(dim=-1).clamp_min(1e-8)).flatten()
    s = e.sort().values
    q = lambda f: float(s[min(int(f * s.numel()), s.numel() - 1)])   # noqa: E731
    return dict(max=float(s[-1]), p9999=q(0.9999), p99=q(0.99), mean=float(e.mean()))


def load_oracle(path):
    """The reference's load_model protocol onto the CPU oracle: returns (oracle, constructor kwargs, load result)."""
    from dust3r_amd.model import parse_model_string
    from oracle.dust3r_ref import DUSt3RRef
    if os.path.isdir(path):         # a hub snapshot directory (config.json + model.safetensors | pytorch_model.bin): what PyTorchModelHubMixin.from_pretrained reads
        import json
        with open(os.path.join(path, 'config.json')) as f:
            kw = {k: (tuple(v) if isinstance(v, list) else v) for k, v in json.load(f).items()}
        for k in ('landscape_only', 'patch_embed_cls', 'freeze'):
            kw.pop(k, None)
        if os.path.isfile(os.path.join(path, 'model.safetensors')):
            from safetensors.torch import load_file
            state = load_file(os.path.join(path, 'model.safetensors'), device='cpu')
        else:
            state = torch.load(os.path.join(path, 'pytorch_model.bin'), map_location='cpu', weights_only=True)
        oracle = DUSt3RRef(**kw).eval()
        # safetensors keeps ONE name per shared tensor (the DPT head registers scratch.layer{i+1}_rn also as scratch.layer_rn[i]): fill the twin of every such
        # key from its alias, as the engine's loader does, so that "missing" below means missing
        import re
        for k in list(state):
            m = re.search(r'scratch\.layer_rn\.(\d)\.', k)
            if m:
                state.setdefault(k.replace(f'scratch.layer_rn.{m.group(1)}.', f'scratch.layer{int(m.group(1)) + 1}_rn.'), state[k])
            m = re.search(r'scratch\.layer(\d)_rn\.', k)
            if m:
                state.setdefault(k.replace(f'scratch.layer{m.group(1)}_rn.', f'scratch.layer_rn.{int(m.group(1)) - 1}.'), state[k])
        if not any(k.startswith('dec_blocks2') for k in state):        # dust3r/model.py:91-98
            for k, v in list(state.items()):
                if k.startswith('dec_blocks'):
                    state[k.replace('dec_blocks', 'dec_blocks2')] = v
        res = oracle.load_state_dict(state, strict=False)
        return oracle, kw, res
    ckpt = torch.load(path, map_location='cpu', weights_only=False)
    args = ckpt['args'].model if hasattr(ckpt['args'], 'model') else ckpt['args']['model']
    kw = parse_model_string(args.replace('ManyAR_PatchEmbed', 'PatchEmbedDust3R'))
    kw.pop('landscape_only', None)
    kw.pop('patch_embed_cls', None)
    kw.pop('freeze', None)
    oracle = DUSt3RRef(**kw).eval()
    state = dict(ckpt['model'])
    if not any(k.startswith('dec_blocks2') for k in state):        # dust3r/model.py:91-98
        for k, v in list(state.items()):
            if k.startswith('dec_blocks'):
                state[k.replace('dec_blocks', 'dec_blocks2')] = v
    res = oracle.load_state_dict(state, strict=False)
    return oracle, kw, res


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument('checkpoint')
    ap.add_argument('--size', default='512x384', help='WxH of the synthetic images (multiples of 16)')
    ap.add_argument('--pairs', type=int, default=2)
    ap.add_argument('--images', nargs='*', help='image files instead of synthetic images (consecutive files form the pairs)')
    ap.add_argument('--precision', default='fp16x3,fp32')
    ap.add_argument('--align', action='store_true')
    a = ap.parse_args()
    from dust3r_amd import _lib
    from dust3r_amd.model import load_model
    from dust3r_amd.synthetic import synthetic_views
    from oracle import tune_threads
    _lib.require_device()
    tune_threads()
    dev = torch.device('cuda:0')
    oracle, kw, res = load_oracle(a.checkpoint)
    print(f'oracle   <- {a.checkpoint}: constructor {kw}')
    print(f'           missing keys {list(res.missing_keys)[:8]}{" ..." if len(res.missing_keys) > 8 else ""} ({len(res.missing_keys)}), '
          f'unexpected {list(res.unexpected_keys)[:8]}{" ..." if len(res.unexpected_keys) > 8 else ""} ({len(res.unexpected_keys)})')
    keys_ok = len(res.missing_keys) == 0 and len(res.unexpected_keys) == 0      # a key-for-key match of the restated module tree is part of the verdict
    if not keys_ok:
        print('           [FAIL] the restated module tree does not match the checkpoint key for key')
    if os.path.isdir(a.checkpoint):
        from dust3r_amd.model import AsymmetricCroCo3DStereo
        engine = AsymmetricCroCo3DStereo.from_pretrained(a.checkpoint, precision=a.precision.split(',')[0]).to(dev)
    else:
        engine = load_model(a.checkpoint, dev, verbose=False, precision=a.precision.split(',')[0])
    print(f'engine   <- loaded, {engine.device_bytes() / 2**30:.2f} GiB in HBM, head {engine.head_type}, depth_mode {engine.depth_mode}, conf_mode {engine.conf_mode}')
    if a.images:
        from dust3r_amd.utils.image import load_images
        W = int(a.size.split('x')[0])
        views = load_images(a.images, size=W, verbose=False)
        v1 = dict(img=torch.cat([v['img'] for v in views[0::2]]), true_shape=torch.cat([torch.as_tensor(v['true_shape']) for v in views[0::2]]), idx=[0] * (len(views) // 2), instance=['0'] * (len(views) // 2))
        v2 = dict(img=torch.cat([v['img'] for v in views[1::2]]), true_shape=torch.cat([torch.as_tensor(v['true_shape']) for v in views[1::2]]), idx=[1] * (len(views) // 2), instance=['1'] * (len(views) // 2))
    else:
        W, H = (int(x) for x in a.size.split('x'))
        v1, v2 = synthetic_views(a.pairs, H, W, seed=0)
    with torch.no_grad():
        r1, r2 = oracle(v1, v2)
    ref = torch.cat((r1['pts3d'], r2['pts3d_in_other_view']))
    nrm = ref.norm(dim=-1)
    print(f'oracle pointmaps: |pts| mean {float(nrm.mean()):.3f}, min / mean {float(nrm.min() / nrm.mean()):.2e}; conf range [{float(r1["conf"].min()):.2f}, {float(r1["conf"].max()):.2f}]')
    ok = keys_ok
    for prec in a.precision.split(','):
        engine.set_precision(prec)
        e1, e2 = engine(v1, v2)
        s = stats(torch.cat((e1['pts3d'], e2['pts3d_in_other_view'])), ref)
        cref = torch.cat((r1['conf'], r2['conf']))
        cerr = float(((torch.cat((e1['conf'], e2['conf'])).cpu() - cref).abs() / cref.abs().clamp_min(1e-6)).max())
        verdict = 'PASS' if s['max'] <= 1e-3 and cerr <= 3e-3 else 'FAIL'          # confidences: the bound the test-suite holds them to
        ok = ok and s['max'] <= 1e-3 and cerr <= 3e-3
        print(f'engine {prec:7s} vs oracle: pointmap rel err max {s["max"]:.3e}  p99.99 {s["p9999"]:.3e}  p99 {s["p99"]:.3e}  mean {s["mean"]:.3e};  conf rel err max {cerr:.3e}   [{verdict} at 1e-3 / conf 3e-3]')
    if a.align:       # needs weights that produce a scene (a real checkpoint): the MST / focal initialisation of a random network's output ends in NaN
      try:
        from dust3r_amd.cloud_opt import GlobalAlignerMode, global_aligner
        from dust3r_amd.image_pairs import make_pairs
        from dust3r_amd.inference import inference
        from oracle.aligner_ref import AlignerRef
        engine.set_precision(a.precision.split(',')[0])
        imgs = [dict(img=v['img'][i:i + 1], true_shape=torch.as_tensor(v['true_shape'])[i:i + 1].numpy(), idx=2 * i + s, instance=str(2 * i + s))
                for i in range(v1['img'].shape[0]) for s, v in ((0, v1), (1, v2))]
        out = inference(make_pairs(imgs, 'complete', None, True), engine, dev, batch_size=8, verbose=False)
        scene = global_aligner(out, dev, mode=GlobalAlignerMode.PointCloudOptimizer, verbose=False)
        from dust3r_amd.cloud_opt import init_im_poses
        init_im_poses.init_minimum_spanning_tree(scene, niter_PnP=10)
        init = {k: v.detach().cpu().clone() for k, v in scene.state_dict(trainable=True).items()}
        ref_al = AlignerRef(out).load_state(init)
        ref_losses = ref_al.run(niter=30, total=300)
        from dust3r_amd.cloud_opt.base_opt import global_alignment_loop
        hist = [global_alignment_loop(scene, niter=1, schedule='cosine', lr=0.01) for _ in range(1)]
        scene.load_state_dict(init)
        loss = scene.compute_global_alignment(init=None, niter=300, schedule='cosine', lr=0.01)
        print(f'aligner: oracle loss after 30 of 300 iterations {ref_losses[-1]:.6f} (first {ref_losses[0]:.6f}); engine first-iteration loss {hist[0]:.6f}, after 300 iterations {loss:.6f}')
      except Exception as e:      # the forward verdict above stands on its own
        print(f'aligner: leg failed: {e!r}')
    sys.exit(0 if ok else 1)


if __name__ == '__main__':
    main()
