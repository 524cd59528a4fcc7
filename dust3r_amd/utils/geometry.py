"""Geometry helpers -- mirror of the reference `dust3r/utils/geometry.py` entries on the hot path:
`xy_grid` (:15-37), `geotrf` (:40-101), `inv` (:104-111), `depthmap_to_camera_coordinates` /
`depthmap_to_absolute_camera_coordinates` (:165-220), `get_med_dist_between_poses` (:364-366).
"""
import numpy as np
import torch


def xy_grid(W, H, device=None, origin=(0, 0), unsqueeze=None, cat_dim=-1, homogeneous=False, **arange_kw):
    """(H, W, 2) grid with out[j, i] = (i + origin[0], j + origin[1]); numpy when device is None."""
    if device is None:
        tw, th = [np.arange(o, o + s, **arange_kw) for s, o in zip((W, H), origin)]
        grid = tuple(np.meshgrid(tw, th, indexing='xy'))
        if homogeneous:
            grid = grid + (np.ones((H, W)),)
        if unsqueeze is not None:
            grid = tuple(np.expand_dims(g, unsqueeze) for g in grid)
        return np.stack(grid, cat_dim) if cat_dim is not None else grid
    tw, th = [torch.arange(o, o + s, device=device, **arange_kw) for s, o in zip((W, H), origin)]
    grid = tuple(torch.meshgrid(tw, th, indexing='xy'))
    if homogeneous:
        grid = grid + (torch.ones((H, W), device=device),)
    if unsqueeze is not None:
        grid = tuple(g.unsqueeze(unsqueeze) for g in grid)
    return torch.stack(grid, cat_dim) if cat_dim is not None else grid


def geotrf(Trf, pts, ncol=None, norm=False):
    """Apply (batched) 3x3 / 4x4 transforms to (..., 2|3) points; `norm` projects on the z=norm plane."""
    assert Trf.ndim >= 2
    if isinstance(Trf, np.ndarray):
        pts = np.asarray(pts)
    elif isinstance(Trf, torch.Tensor):
        pts = torch.as_tensor(pts, dtype=Trf.dtype)
    out_shape = pts.shape[:-1]
    ncol = ncol or pts.shape[-1]

    if isinstance(Trf, torch.Tensor) and isinstance(pts, torch.Tensor) and Trf.ndim == 3 and pts.ndim == 4:
        d = pts.shape[3]
        if Trf.shape[-1] == d:
            pts = torch.einsum('bij, bhwj -> bhwi', Trf, pts)
        elif Trf.shape[-1] == d + 1:
            pts = torch.einsum('bij, bhwj -> bhwi', Trf[:, :d, :d], pts) + Trf[:, None, None, :d, d]
        else:
            raise ValueError(f'bad shape, not ending with 3 or 4, for {pts.shape=}')
    else:
        if Trf.ndim >= 3:
            n = Trf.ndim - 2
            assert Trf.shape[:n] == pts.shape[:n], 'batch size does not match'
            Trf = Trf.reshape(-1, Trf.shape[-2], Trf.shape[-1])
            if pts.ndim > Trf.ndim:
                pts = pts.reshape(Trf.shape[0], -1, pts.shape[-1])
            elif pts.ndim == 2:
                pts = pts[:, None, :]
        if pts.shape[-1] + 1 == Trf.shape[-1]:
            Trf = Trf.swapaxes(-1, -2)
            pts = pts @ Trf[..., :-1, :] + Trf[..., -1:, :]
        elif pts.shape[-1] == Trf.shape[-1]:
            Trf = Trf.swapaxes(-1, -2)
            pts = pts @ Trf
        else:
            pts = Trf @ pts.T
            if pts.ndim >= 2:
                pts = pts.swapaxes(-1, -2)
    if norm:
        pts = pts / pts[..., -1:]
        if norm != 1:
            pts *= norm
    return pts[..., :ncol].reshape(*out_shape, ncol)


def inv(mat):
    if isinstance(mat, torch.Tensor):
        return torch.linalg.inv(mat)
    if isinstance(mat, np.ndarray):
        return np.linalg.inv(mat)
    raise ValueError(f'bad matrix type = {type(mat)}')


def depthmap_to_camera_coordinates(depthmap, camera_intrinsics, pseudo_focal=None):
    camera_intrinsics = np.float32(camera_intrinsics)
    H, W = depthmap.shape
    assert camera_intrinsics[0, 1] == 0.0 and camera_intrinsics[1, 0] == 0.0
    if pseudo_focal is None:
        fu, fv = camera_intrinsics[0, 0], camera_intrinsics[1, 1]
    else:
        assert pseudo_focal.shape == (H, W)
        fu = fv = pseudo_focal
    cu, cv = camera_intrinsics[0, 2], camera_intrinsics[1, 2]
    u, v = np.meshgrid(np.arange(W), np.arange(H))
    z_cam = depthmap
    x_cam = (u - cu) * z_cam / fu
    y_cam = (v - cv) * z_cam / fv
    X_cam = np.stack((x_cam, y_cam, z_cam), axis=-1).astype(np.float32)
    return X_cam, (depthmap > 0.0)


def depthmap_to_absolute_camera_coordinates(depthmap, camera_intrinsics, camera_pose, **kw):
    X_cam, valid_mask = depthmap_to_camera_coordinates(depthmap, camera_intrinsics)
    X_world = X_cam
    if camera_pose is not None:
        R, t = camera_pose[:3, :3], camera_pose[:3, 3]
        X_world = np.einsum('ik, vuk -> vui', R, X_cam) + t[None, None, :]
    return X_world, valid_mask


def get_med_dist_between_poses(poses):
    from scipy.spatial.distance import pdist
    from .device import to_numpy
    return np.median(pdist([to_numpy(p[:3, 3]) for p in poses]))


def find_reciprocal_matches(P1, P2):
    """Mirror of the reference `find_reciprocal_matches` (dust3r/utils/geometry.py:345-361; caller visloc.py:105): mutual nearest
    neighbours between two 3-D point sets. Returns (reciprocal_in_P2 bool (len P2), nn2_in_P1 int (len P2), number of matches),
    numpy arrays for numpy inputs and torch tensors for torch inputs. The two nearest-neighbour queries run as exhaustive scans
    on the GPU (d3r_nearest_neighbors) instead of SciPy KD-trees; exact distance ties resolve to the lowest index."""
    import ctypes as C

    from .. import _lib
    from .._lib import check, current_stream, lib, ptr
    _lib.require_device()
    as_numpy = isinstance(P1, np.ndarray)
    dev = P1.device if (isinstance(P1, torch.Tensor) and P1.is_cuda) else torch.device('cuda', torch.cuda.current_device())
    a = torch.as_tensor(P1, dtype=torch.float32).reshape(-1, 3).to(dev).contiguous()
    b = torch.as_tensor(P2, dtype=torch.float32).reshape(-1, 3).to(dev).contiguous()
    nn1_in_P2 = torch.empty(len(a), dtype=torch.int32, device=dev)
    nn2_in_P1 = torch.empty(len(b), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        check(lib.d3r_nearest_neighbors(ptr(a), len(a), ptr(b), len(b), ptr(nn1_in_P2), current_stream()), 'nearest_neighbors')
        check(lib.d3r_nearest_neighbors(ptr(b), len(b), ptr(a), len(a), ptr(nn2_in_P1), current_stream()), 'nearest_neighbors')
    nn1, nn2 = nn1_in_P2.long(), nn2_in_P1.long()
    reciprocal_in_P2 = nn1[nn2] == torch.arange(len(nn2), device=dev)
    count = int(reciprocal_in_P2.sum())
    if as_numpy:
        return reciprocal_in_P2.cpu().numpy(), nn2.cpu().numpy(), count
    return reciprocal_in_P2, nn2, count
