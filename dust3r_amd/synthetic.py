"""Synthetic workloads for tests and bench.py (no dataset / checkpoint is reachable):
model configurations, seeded weights addressed by state-dict key, seeded image pairs and
seeded multi-view scenes for the global aligner.
"""
import math
import re
import zlib

import numpy as np
import torch

inf = float('inf')

MODEL_CONFIGS = {
    # README.md:99-103 / 318
    'DUSt3R_ViTLarge_BaseDecoder_512_dpt': dict(pos_embed='RoPE100', img_size=(512, 512), head_type='dpt',
                                                enc_embed_dim=1024, enc_depth=24, enc_num_heads=16,
                                                dec_embed_dim=768, dec_depth=12, dec_num_heads=12),
    'DUSt3R_ViTLarge_BaseDecoder_512_linear': dict(pos_embed='RoPE100', img_size=(512, 512), head_type='linear',
                                                   enc_embed_dim=1024, enc_depth=24, enc_num_heads=16,
                                                   dec_embed_dim=768, dec_depth=12, dec_num_heads=12),
    'DUSt3R_ViTLarge_BaseDecoder_224_linear': dict(pos_embed='RoPE100', img_size=(224, 224), head_type='linear',
                                                   enc_embed_dim=1024, enc_depth=24, enc_num_heads=16,
                                                   dec_embed_dim=768, dec_depth=12, dec_num_heads=12),
    # small shapes with the same topology, for fast parity tests (head dim stays 64)
    'tiny_dpt': dict(pos_embed='RoPE100', img_size=(64, 64), head_type='dpt', enc_embed_dim=256, enc_depth=2,
                     enc_num_heads=4, dec_embed_dim=128, dec_depth=12, dec_num_heads=2),
    'tiny_linear': dict(pos_embed='RoPE100', img_size=(64, 64), head_type='linear', enc_embed_dim=256, enc_depth=2,
                        enc_num_heads=4, dec_embed_dim=128, dec_depth=10, dec_num_heads=2),
}


def synthetic_state_dict(reference_state, seed=0, out_gain=1.0, device='cpu'):
    """Deterministic weights addressed by KEY NAME (not by construction order), so the
    reference, this oracle and the HIP engine can all be filled identically:
    every tensor gets its own generator seeded with crc32(name) ^ seed.
    Matrices/convs: N(0, 1/fan_in) (keeps activations O(1) through the depth); LayerNorm
    weight 1 + 0.1 N; every bias 0.02 N (non-zero on purpose: exercises the bias paths).
    `out_gain` scales the last head layer so that |xyz| is O(1) before expm1.
    `device`: where the tensors are generated ('cpu' values are what the golden fixtures were made with; a CUDA
    device uses that device's generator -- same distributions, different values -- for the full-size bench model,
    whose weights never need to exist on the host)."""
    out = {}
    device = torch.device(device)
    for name in sorted(reference_state.keys()):
        ref = reference_state[name]
        g = torch.Generator(device=device).manual_seed((zlib.crc32(name.encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF)
        shape = tuple(ref.shape)
        if ref.ndim >= 2:
            # ConvTranspose2d weight (act_postprocess.{0,1}.1) is (Cin, Cout, k, k): one tap per output pixel
            fan_in = ref.shape[0] if re.search(r'act_postprocess\.[01]\.1\.weight$', name) else ref[0].numel()
            t = torch.randn(shape, generator=g, device=device) * fan_in ** -0.5
        elif name.endswith('weight'):
            t = 1 + 0.1 * torch.randn(shape, generator=g, device=device)
        else:
            t = 0.02 * torch.randn(shape, generator=g, device=device)
        out[name] = t.to(ref.dtype)
    # aliased tensors (scratch.layerN_rn <-> scratch.layer_rn.N) must stay identical
    for name in list(out):
        if '.scratch.layer_rn.' in name:
            n = int(name.split('.scratch.layer_rn.')[1].split('.')[0])
            out[name.replace(f'.scratch.layer_rn.{n}.', f'.scratch.layer{n + 1}_rn.')] = out[name]
    for name in out:
        if name.endswith('dpt.head.4.weight') or name.endswith('dpt.head.4.bias') or \
                (name.startswith('downstream_head') and '.proj.' in name):
            out[name] = out[name] * out_gain
    return out


# measured with seed 0 on the fp32 oracle: brings the mean pre-activation |xyz| to ~1 (see synthetic_state_dict)
OUT_GAIN = {'DUSt3R_ViTLarge_BaseDecoder_512_dpt': 0.3, 'DUSt3R_ViTLarge_BaseDecoder_512_linear': 0.6,
            'DUSt3R_ViTLarge_BaseDecoder_224_linear': 0.6, 'tiny_dpt': 0.15, 'tiny_linear': 0.6}




def synthetic_views(n_pairs, H, W, seed=0, device='cpu'):
    """img ~ U(-1,1) matching ImgNorm's range (reference dust3r/utils/image.py:23).
    Returns (view1, view2) in the collated view-dict format of dust3r/utils/image.py:122-123."""
    g = torch.Generator().manual_seed(seed)
    img1 = (torch.rand((n_pairs, 3, H, W), generator=g) * 2 - 1).to(device)
    img2 = (torch.rand((n_pairs, 3, H, W), generator=g) * 2 - 1).to(device)
    ts = torch.tensor([[H, W]] * n_pairs, dtype=torch.int32)
    v1 = dict(img=img1, true_shape=ts, idx=list(range(0, 2 * n_pairs, 2)),
              instance=[str(i) for i in range(0, 2 * n_pairs, 2)])
    v2 = dict(img=img2, true_shape=ts.clone(), idx=list(range(1, 2 * n_pairs, 2)),
              instance=[str(i) for i in range(1, 2 * n_pairs, 2)])
    return v1, v2


def synthetic_image_list(n_views, H, W, seed=0):
    """A list of single-view dicts, as `load_images` would return (utils/image.py:122-123)."""
    g = torch.Generator().manual_seed(seed)
    return [dict(img=torch.rand((1, 3, H, W), generator=g) * 2 - 1, true_shape=np.int32([[H, W]]), idx=i,
                 instance=str(i)) for i in range(n_views)]


def synthetic_photo(W, H, seed):
    """Deterministic smooth RGB test picture, uint8 (H, W, 3): gradients, a few sinusoids and mild noise, so that resampling filters act
    non-trivially on it (load_images tests; no image file is shipped)."""
    rng = np.random.RandomState(seed)
    y, x = np.mgrid[:H, :W].astype(np.float64)
    chans = []
    for c in range(3):
        a, b, p1, p2 = rng.uniform(0.01, 0.2, 4)
        v = 0.5 + 0.25 * np.sin(a * x + p1 * 10) * np.cos(b * y + p2 * 10) + 0.2 * (x / W - y / H) * (1 if c != 1 else -1)
        chans.append(v + 0.03 * rng.randn(H, W))
    return (np.clip(np.stack(chans, axis=-1), 0, 1) * 255).round().astype('uint8')


# (source W, H, size, square_ok): landscape / portrait / square (4:3 rule and square_ok) / odd sizes / enlargement (BICUBIC) / the 224 rule
LOAD_IMAGES_CASES = [(640, 480, 160, False), (480, 640, 160, False), (500, 500, 160, False), (500, 500, 160, True), (333, 517, 128, False),
                     (100, 75, 160, False), (417, 300, 224, False), (300, 417, 224, False), (1001, 333, 192, False)]


# ------------------------------------------------------------------ aligner scenes
def _rotmat_to_quat_xyzw(R):
    """numpy, single 3x3 -> XYZW unit quaternion (largest-component branch)."""
    m = np.asarray(R, np.float64)
    dec = np.array([1 + m[0, 0] - m[1, 1] - m[2, 2], 1 - m[0, 0] + m[1, 1] - m[2, 2],
                    1 - m[0, 0] - m[1, 1] + m[2, 2], 1 + m[0, 0] + m[1, 1] + m[2, 2]])
    c = int(dec.argmax())
    if c == 3:
        q = np.array([m[2, 1] - m[1, 2], m[0, 2] - m[2, 0], m[1, 0] - m[0, 1], dec[3]])
    elif c == 0:
        q = np.array([dec[0], m[1, 0] + m[0, 1], m[0, 2] + m[2, 0], m[2, 1] - m[1, 2]])
    elif c == 1:
        q = np.array([m[1, 0] + m[0, 1], dec[1], m[2, 1] + m[1, 2], m[0, 2] - m[2, 0]])
    else:
        q = np.array([m[0, 2] + m[2, 0], m[2, 1] + m[1, 2], dec[2], m[1, 0] - m[0, 1]])
    return q / np.linalg.norm(q)


def _axis_angle_R(axis, angle):
    axis = np.asarray(axis, np.float64)
    axis = axis / np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + math.sin(angle) * K + (1 - math.cos(angle)) * (K @ K)


def _signed_log1p(x):
    return np.sign(x) * np.log1p(np.abs(x))


def scene_edges(n_views, scene_graph='complete', symmetrize=False):
    """Edge list in the order `make_pairs` produces (reference dust3r/image_pairs.py:11-68)."""
    from dust3r_amd.image_pairs import make_pairs
    stubs = [dict(idx=i) for i in range(n_views)]
    return [(a['idx'], b['idx']) for a, b in make_pairs(stubs, scene_graph, None, symmetrize)]


def synthetic_scene(n_views, H, W, seed=0, scene_graph='complete', symmetrize=False, noise=0.01, device='cpu',
                    perturb=True, device_rng=False):
    """Multi-view scene for the global aligner: cameras on a circle of radius 2
    looking at the origin, focal 1.2 W, depth 2 + 0.5 smooth-noise; per edge the pairwise
    pointmaps are the exact geometry times a random scale U(0.5, 2) plus N(0, noise), conf =
    1 + exp(N(1, 0.5)). Returns (dust3r_output, init_state, gt):
      dust3r_output  dict(view1, view2, pred1, pred2) exactly as `inference()` returns it
      init_state     trainable state (pw_poses, im_poses, im_depthmaps, im_focals) = ground truth
                     perturbed (rotations <=5 deg, translations 10 %, log-depth N(0, .05)), to be
                     loaded into oracle and engine alike (no RNG inside the optimisation loop)
      gt             cam2world (n,4,4), focal, depth (n,H,W)
    `device_rng`: draw the per-edge noise / confidences with `device`'s generator instead of the CPU one (same distributions, other values;
    600 edges at 512x384 take 1 s instead of 30 s) -- for timing workloads only: the golden fixtures and parity tests use the CPU values.
    """
    rng = np.random.RandomState(seed)
    g = torch.Generator(device='cpu').manual_seed(seed)
    A = H * W
    f_gt = 1.2 * W
    # ground truth cameras
    c2w = np.zeros((n_views, 4, 4))
    for k in range(n_views):
        a = 2 * math.pi * k / n_views
        c = np.array([2 * math.cos(a), 0.3 * math.sin(2 * a), 2 * math.sin(a)])
        z = -c / np.linalg.norm(c)
        x = np.cross(np.array([0., 1., 0.]), z)
        x /= np.linalg.norm(x)
        y = np.cross(z, x)
        c2w[k, :3, 0], c2w[k, :3, 1], c2w[k, :3, 2], c2w[k, :3, 3], c2w[k, 3, 3] = x, y, z, c, 1
    c2w_t = torch.tensor(c2w, dtype=torch.float32, device=device)
    # ground truth depth: low-res noise, bilinearly upsampled
    low = torch.randn((n_views, 1, 6, 8), generator=g).clamp(-2, 2)
    depth = 2 + 0.25 * torch.nn.functional.interpolate(low, size=(H, W), mode='bilinear', align_corners=True)[:, 0]
    depth = depth.to(device)
    vs, us = torch.meshgrid(torch.arange(H, device=device, dtype=torch.float32),
                            torch.arange(W, device=device, dtype=torch.float32), indexing='ij')
    rays = torch.stack(((us - W / 2) / f_gt, (vs - H / 2) / f_gt, torch.ones_like(us)), dim=-1)     # (H,W,3)
    cam_pts = depth[..., None] * rays[None]                                                      # (n,H,W,3)
    world = torch.einsum('nij,nhwj->nhwi', c2w_t[:, :3, :3], cam_pts) + c2w_t[:, None, None, :3, 3]

    edges = scene_edges(n_views, scene_graph, symmetrize)
    E = len(edges)
    scales = rng.uniform(0.5, 2.0, size=E)
    w2c = torch.linalg.inv(c2w_t)
    pred_i = torch.empty((E, H, W, 3), dtype=torch.float32, device=device)
    pred_j = torch.empty((E, H, W, 3), dtype=torch.float32, device=device)
    conf_i = torch.empty((E, H, W), dtype=torch.float32, device=device)
    conf_j = torch.empty((E, H, W), dtype=torch.float32, device=device)
    gd = torch.Generator(device=device).manual_seed(seed) if device_rng else None
    rnd = (lambda shape: torch.randn(shape, generator=gd, device=device)) if device_rng else (lambda shape: torch.randn(shape, generator=g).to(device))
    for e, (i, j) in enumerate(edges):
        s = float(scales[e])
        pj = torch.einsum('ij,hwj->hwi', w2c[i, :3, :3], world[j]) + w2c[i, :3, 3]
        pred_i[e] = s * cam_pts[i] + noise * rnd((H, W, 3))
        pred_j[e] = s * pj + noise * rnd((H, W, 3))
        conf_i[e] = 1 + torch.exp(1 + 0.5 * rnd((H, W)))
        conf_j[e] = 1 + torch.exp(1 + 0.5 * rnd((H, W)))
    ts = torch.tensor([[H, W]] * E, dtype=torch.int32)
    output = dict(
        view1=dict(idx=[i for i, j in edges], instance=[str(i) for i, j in edges], true_shape=ts),
        view2=dict(idx=[j for i, j in edges], instance=[str(j) for i, j in edges], true_shape=ts.clone()),
        pred1=dict(pts3d=pred_i, conf=conf_i),
        pred2=dict(pts3d_in_other_view=pred_j, conf=conf_j), loss=None)

    # initial trainable state in the optimiser's parameterisation (reference base_opt.py:157-176,
    # optimizer.py:121-125,156-162): world rescaled so that mean_e log(pw scale) == log(base_scale=0.5)
    cworld = 0.5 * math.exp(float(np.mean(np.log(scales))))
    pw = np.zeros((E, 8), np.float32)
    for e, (i, j) in enumerate(edges):
        R = c2w[i, :3, :3]
        T = cworld * c2w[i, :3, 3]
        sc = cworld / scales[e]
        if perturb:
            R = _axis_angle_R(rng.randn(3), np.deg2rad(rng.uniform(-5, 5))) @ R
            T = T * (1 + 0.1 * rng.uniform(-1, 1, size=3))
            sc = sc * math.exp(0.05 * rng.randn())
        pw[e, :4] = _rotmat_to_quat_xyzw(R)
        pw[e, 4:7] = _signed_log1p(T / sc)
        pw[e, 7] = math.log(sc)
    imp = np.zeros((n_views, 7), np.float32)
    for k in range(n_views):
        R = c2w[k, :3, :3]
        T = cworld * c2w[k, :3, 3]
        if perturb:
            R = _axis_angle_R(rng.randn(3), np.deg2rad(rng.uniform(-5, 5))) @ R
            T = T * (1 + 0.1 * rng.uniform(-1, 1, size=3))
        imp[k, :4] = _rotmat_to_quat_xyzw(R)
        imp[k, 4:7] = _signed_log1p(T)
    logd = torch.log(cworld * depth).reshape(n_views, A).cpu()
    if perturb:
        logd = logd + 0.05 * torch.randn(logd.shape, generator=g)
    focal0 = f_gt * (1.1 if perturb else 1.0)
    init_state = dict(pw_poses=torch.from_numpy(pw), pw_adaptors=torch.zeros((E, 2)),
                      im_poses=torch.from_numpy(imp), im_depthmaps=logd.float().contiguous(),
                      im_focals=torch.full((n_views, 1), 20 * math.log(focal0), dtype=torch.float32),
                      im_pp=torch.zeros((n_views, 2)))
    gt = dict(cam2world=c2w_t.cpu(), focal=f_gt, depth=depth.cpu(), world_scale=cworld, edges=edges,
              edge_scales=scales)
    return output, init_state, gt
