"""The demo's reconstruction body on the engine: the call sequence of the reference's `get_reconstructed_scene` +
`get_3D_model_from_scene` (dust3r/demo.py:110-186, minus gradio / GLB export / matplotlib), issued against dust3r_amd under the
reference's names -- files on disk -> load_images -> make_pairs -> inference -> global_aligner -> compute_global_alignment(init='mst')
-> clean_pointcloud -> getters. The network has random weights (no checkpoint is reachable), so its pointmaps carry no geometry: what is
checked is that every call of the body runs on the engine and returns the reference's structures -- the numbers of each stage are pinned
by the parity tests (forward: test_forward_gpu.py; alignment on consistent scenes: test_aligner_gpu.py). (tests/test_oracle_pins.py checks in the build container that the reference's own demo module binds
to these functions through the INTEGRATION.md aliasing.)"""
import copy
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _write_images(tmp_path, sizes):
    import PIL.Image
    from dust3r_amd.synthetic import synthetic_photo
    paths = []
    for k, (W, H) in enumerate(sizes):
        p = os.path.join(str(tmp_path), f'view{k}.png')
        PIL.Image.fromarray(synthetic_photo(W, H, seed=20 + k)).save(p)
        paths.append(p)
    return paths


def _engine(gpu):
    from dust3r_amd.model import AsymmetricCroCo3DStereo
    from dust3r_amd.synthetic import MODEL_CONFIGS
    from oracle.dust3r_ref import build_ref_model
    m = AsymmetricCroCo3DStereo(landscape_only=False, **MODEL_CONFIGS['tiny_dpt'])        # default precision = the parity-grade mode
    assert m.precision == 'fp16x3'
    m.load_state_dict(build_ref_model('tiny_dpt').state_dict())
    return m.to(gpu)


def reconstruct(filelist, model, device, image_size, schedule, niter, min_conf_thr, clean_depth, scenegraph_type, winsize=1, refid=0):
    """dust3r/demo.py:135-186 + :110-132, same order, same arguments."""
    from dust3r_amd.cloud_opt import GlobalAlignerMode, global_aligner
    from dust3r_amd.image_pairs import make_pairs
    from dust3r_amd.inference import inference
    from dust3r_amd.utils.device import to_numpy
    from dust3r_amd.utils.image import load_images
    try:
        square_ok = model.square_ok
    except Exception:
        square_ok = False
    imgs = load_images(filelist, size=image_size, verbose=False, patch_size=model.patch_size, square_ok=square_ok)
    if len(imgs) == 1:
        imgs = [imgs[0], copy.deepcopy(imgs[0])]
        imgs[1]['idx'] = 1
    if scenegraph_type == 'swin':
        scenegraph_type = scenegraph_type + '-' + str(winsize)
    elif scenegraph_type == 'oneref':
        scenegraph_type = scenegraph_type + '-' + str(refid)
    pairs = make_pairs(imgs, scene_graph=scenegraph_type, prefilter=None, symmetrize=True)
    output = inference(pairs, model, device, batch_size=1, verbose=False)
    mode = GlobalAlignerMode.PointCloudOptimizer if len(imgs) > 2 else GlobalAlignerMode.PairViewer
    scene = global_aligner(output, device=device, mode=mode, verbose=False)
    loss = None
    if mode == GlobalAlignerMode.PointCloudOptimizer:
        loss = scene.compute_global_alignment(init='mst', niter=niter, schedule=schedule, lr=0.01)
    if clean_depth:
        scene = scene.clean_pointcloud()
    rgbimg = scene.imgs
    focals = scene.get_focals().cpu()
    cams2world = scene.get_im_poses().cpu()
    pts3d = to_numpy(scene.get_pts3d())
    scene.min_conf_thr = float(scene.conf_trf(torch.tensor(min_conf_thr)))
    msk = to_numpy(scene.get_masks())
    depths = to_numpy(scene.get_depthmaps())
    confs = to_numpy([c for c in scene.im_conf])
    return scene, loss, dict(rgbimg=rgbimg, focals=focals, cams2world=cams2world, pts3d=pts3d, msk=msk, depths=depths, confs=confs, output=output)


@pytest.mark.parametrize('graph,n', [('complete', 3), ('swin', 4), ('oneref', 3)])
def test_demo_body_multi_view(gpu, tmp_path, graph, n):
    files = _write_images(tmp_path, [(200, 150)] * n)
    model = _engine(gpu)
    scene, loss, out = reconstruct(files, model, gpu, image_size=96, schedule='linear', niter=30, min_conf_thr=3.0, clean_depth=True,
                                   scenegraph_type=graph)
    H, W = 64, 96                                     # 200x150 -> long side 96 -> 96x72 -> cropped to multiples of 16
    assert isinstance(loss, float)            # (a random-weight network gives degenerate focals: the loss itself may be nan, as in the reference)
    o = out['output']
    assert o['pred1']['pts3d'].shape == (n * (n - 1) if graph == 'complete' else len(o['view1']['idx']), H, W, 3) and torch.isfinite(o['pred1']['pts3d']).all()
    assert o['pred1']['conf'].min() >= 1 and o['view1']['img'].shape[1:] == (3, H, W) and not o['pred1']['pts3d'].is_cuda
    assert len(out['rgbimg']) == n and out['rgbimg'][0].shape == (H, W, 3) and 0 <= out['rgbimg'][0].min() and out['rgbimg'][0].max() <= 1
    assert out['focals'].shape == (n, 1) and out['cams2world'].shape == (n, 4, 4)
    assert len(out['pts3d']) == n and out['pts3d'][0].shape == (H, W, 3) and out['msk'][0].shape == (H, W) and out['msk'][0].dtype == bool
    assert out['depths'][0].shape == (H, W) and out['confs'][0].shape == (H, W)


def test_demo_body_single_image_and_pair(gpu, tmp_path):
    """One file (the demo duplicates it) and two files: both go through GlobalAlignerMode.PairViewer."""
    model = _engine(gpu)
    for sizes in ([(160, 160)], [(200, 150), (180, 150)]):
        files = _write_images(tmp_path, sizes)
        # size 128: a square picture becomes 128 x 96 (the 4:3 rule), both multiples of the patch size
        scene, loss, out = reconstruct(files, model, gpu, image_size=128, schedule='linear', niter=10, min_conf_thr=3.0, clean_depth=False,
                                       scenegraph_type='complete')
        assert loss is None and type(scene).__name__ == 'PairViewer'
        assert len(out['pts3d']) == 2 and out['cams2world'].shape == (2, 4, 4) and out['focals'].shape == (2,)
        assert out['pts3d'][0].shape[:2] == out['msk'][0].shape and out['pts3d'][0].shape[2] == 3


def test_demo_body_mixed_portrait_and_landscape(gpu, tmp_path):
    """A folder mixing landscape and portrait pictures: inference() takes its mixed-shape path (lists), the engine the two-size forward,
    the aligner the padded (max_area) layout."""
    files = _write_images(tmp_path, [(200, 150), (150, 200), (200, 150)])
    model = _engine(gpu)
    scene, loss, out = reconstruct(files, model, gpu, image_size=96, schedule='cosine', niter=20, min_conf_thr=3.0, clean_depth=True,
                                   scenegraph_type='complete')
    assert isinstance(loss, float) and [p.shape[:2] for p in out['pts3d']] == [(64, 96), (96, 64), (64, 96)]
    assert isinstance(out['output']['pred1']['pts3d'], list) and all(torch.isfinite(t).all() for t in out['output']['pred1']['pts3d'])


def test_validate_checkpoint_tool_on_a_synthetic_checkpoint(gpu, tmp_path):
    """tools/validate_checkpoint.py -- the one-command engine-vs-oracle harness for the day a real checkpoint is at hand -- run end to end on a
    checkpoint FILE in the reference's format ({'args': Namespace(model="AsymmetricCroCo3DStereo(...)"), 'model': state_dict}, dust3r/model.py:27-43)
    written from the seeded tiny DPT model: both loads report no missing / unexpected keys, every requested precision passes the 1e-3 bar
    (exit code 0). (The --align legs need weights that produce a scene.)"""
    import argparse
    import subprocess
    import sys
    import torch
    from dust3r_amd.synthetic import MODEL_CONFIGS
    from oracle.dust3r_ref import build_ref_model
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = MODEL_CONFIGS['tiny_dpt']
    oracle = build_ref_model('tiny_dpt')
    state = {k: v for k, v in oracle.state_dict().items() if not k.startswith('dec_blocks2')}      # released checkpoints carry dec_blocks2 too; the loader duplicates when absent
    model_str = ("AsymmetricCroCo3DStereo(pos_embed='RoPE100', patch_embed_cls='ManyAR_PatchEmbed', img_size=(64, 64), head_type='dpt', output_mode='pts3d', "
                 "depth_mode=('exp', -inf, inf), conf_mode=('exp', 1, inf), enc_embed_dim=%d, enc_depth=%d, enc_num_heads=%d, dec_embed_dim=%d, dec_depth=%d, dec_num_heads=%d)"
                 % (cfg['enc_embed_dim'], cfg['enc_depth'], cfg['enc_num_heads'], cfg['dec_embed_dim'], cfg['dec_depth'], cfg['dec_num_heads']))
    path = str(tmp_path / 'tiny_dpt.pth')
    torch.save({'args': argparse.Namespace(model=model_str), 'model': state}, path)
    r = subprocess.run([sys.executable, os.path.join(root, 'tools', 'validate_checkpoint.py'), path, '--size', '96x64', '--pairs', '2', '--precision', 'fp16x3,fp32'],
                       capture_output=True, text=True, timeout=600, cwd=root)
    print(r.stdout[-3000:])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert 'missing keys [] (0), unexpected [] (0)' in r.stdout
    assert r.stdout.count('[PASS at 1e-3 / conf 3e-3]') == 2


def test_validate_checkpoint_tool_on_a_synthetic_hub_snapshot(gpu, tmp_path):
    """The same harness on a hub SNAPSHOT DIRECTORY (what PyTorchModelHubMixin.from_pretrained reads, dust3r/model.py:76-85; `DUST3R_CKPT=<dir> python bench.py` runs exactly
    this first): config.json with lists / Infinity / ManyAR patch embed / freeze, model.safetensors with ONE name per shared tensor (the DPT head's layer_rn convolutions)
    and no dec_blocks2 keys. Oracle and engine both load it key for key and every precision passes the bar."""
    import json
    import re
    import subprocess
    import sys
    from safetensors.torch import save_file
    from dust3r_amd.synthetic import MODEL_CONFIGS
    from oracle.dust3r_ref import build_ref_model
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = dict(MODEL_CONFIGS['tiny_dpt'])
    oracle = build_ref_model('tiny_dpt')
    state = {k: v.detach().contiguous().clone() for k, v in oracle.state_dict().items() if not re.search(r'scratch\.layer\d_rn\.', k) and not k.startswith('dec_blocks2')}
    snap = tmp_path / 'snapshot'
    snap.mkdir()
    cfg_json = {k: (list(v) if isinstance(v, tuple) else v) for k, v in cfg.items()}
    cfg_json.update(landscape_only=False, patch_embed_cls='ManyAR_PatchEmbed', freeze='none', depth_mode=['exp', float('-inf'), float('inf')], conf_mode=['exp', 1, float('inf')])
    (snap / 'config.json').write_text(json.dumps(cfg_json))
    save_file(state, str(snap / 'model.safetensors'))
    r = subprocess.run([sys.executable, os.path.join(root, 'tools', 'validate_checkpoint.py'), str(snap), '--size', '96x64', '--pairs', '1', '--precision', 'fp16x3,fp32'],
                       capture_output=True, text=True, timeout=600, cwd=root)
    print(r.stdout[-3000:])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert 'missing keys [] (0), unexpected [] (0)' in r.stdout
    assert r.stdout.count('[PASS at 1e-3 / conf 3e-3]') == 2
