"""ORACLE (test infrastructure only) -- CPU restatements of the small host routines around the global aligner that the product
evaluates on the GPU (csrc/bootstrap.hip, aligner.hip: d3r_weiszfeld_focals, d3r_clean_pointcloud).

Follows (citations into /root/reference/dust3r/):
  post_process.py:40-56      estimate_focal_knowing_depth(focal_mode='weiszfeld'): closed-form l2 start, then 10 rounds of
                             inverse-distance re-weighting; principal point = image centre in every call site of the hot path
                             (cloud_opt/init_im_poses.py:235-241, pair_viewer.py:44-46)
  cloud_opt/base_opt.py:369-405  clean_pointcloud: sequential over images i, each i against every j != i
Pinned against the unmodified reference functions by tests/test_oracle_pins.py (build container only).
"""
import numpy as np
import torch


def estimate_focal_weiszfeld(pts3d, iterations=10):
    """pts3d (H, W, 3) torch/numpy -> float focal (fp32 arithmetic like the reference, numpy)."""
    p = np.asarray(pts3d.detach().cpu() if isinstance(pts3d, torch.Tensor) else pts3d, np.float32)
    H, W, _ = p.shape
    v, u = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing='ij')
    px = np.stack((u - np.float32(W / 2), v - np.float32(H / 2)), axis=-1).reshape(-1, 2)
    p = p.reshape(-1, 3)
    with np.errstate(divide='ignore', invalid='ignore'):
        r = p[:, :2] / p[:, 2:3]
    r = np.where(np.isfinite(r), r, np.float32(0))           # nan_to_num(posinf=0, neginf=0) (nan -> 0 is the default)
    dot_px, dot_rr = (r * px).sum(-1), (r * r).sum(-1)
    f = dot_px.mean(dtype=np.float32) / dot_rr.mean(dtype=np.float32)
    for _ in range(iterations):
        dist = np.linalg.norm(px - f * r, axis=-1)
        w = np.float32(1) / np.maximum(dist, np.float32(1e-8))
        f = (w * dot_px).mean(dtype=np.float32) / (w * dot_rr).mean(dtype=np.float32)
    return float(max(f, 0.0))                                  # clip(min=0 * base, max=inf)


def clean_pointcloud_ref(confs, K, world2cam, depthmaps, pts3d, tol=0.001, bad_conf=0):
    """confs / depthmaps: lists of (H, W); pts3d: list of (H, W, 3) world points; K (n,3,3); world2cam (n,4,4). Returns the new
    confidences. A point of image i that lands, in camera j, in front of j's depth (by more than tol) on a pixel that is MORE
    confident than the point itself is clipped to bad_conf; image i sees the already cleaned confidences of images < i."""
    n = len(confs)
    res = [(c.detach().cpu().numpy() if isinstance(c, torch.Tensor) else np.asarray(c)).astype(np.float32, copy=True) for c in confs]
    to_np = lambda t: np.asarray(t.detach().cpu() if isinstance(t, torch.Tensor) else t, np.float32)  # noqa: E731
    K, world2cam = to_np(K), to_np(world2cam)
    depthmaps = [to_np(d).reshape(res[i].shape) for i, d in enumerate(depthmaps)]
    pts3d = [to_np(p).reshape(res[i].shape + (3,)) for i, p in enumerate(pts3d)]
    for i in range(n):
        for j in range(n):
            if i == j:
                continue
            cam = pts3d[i] @ world2cam[j, :3, :3].T + world2cam[j, :3, 3]
            z = cam[..., 2]
            img = cam @ K[j].T
            with np.errstate(divide='ignore', invalid='ignore'):
                uv = img[..., :2] / img[..., 2:3]
            uv = np.rint(uv)                                    # torch.round: half to even
            Hj, Wj = res[j].shape
            ok = (z > 0) & np.isfinite(uv).all(-1) & (uv[..., 0] >= 0) & (uv[..., 0] < Wj) & (uv[..., 1] >= 0) & (uv[..., 1] < Hj)
            uu, vv = uv[..., 0][ok].astype(np.int64), uv[..., 1][ok].astype(np.int64)
            bad = (z[ok] < np.float32(1 - tol) * depthmaps[j][vv, uu]) & (res[i][ok] < res[j][vv, uu])
            sel = np.zeros_like(ok)
            sel[ok] = bad
            res[i][sel] = np.minimum(res[i][sel], np.float32(bad_conf))
    return [torch.from_numpy(r) for r in res]
