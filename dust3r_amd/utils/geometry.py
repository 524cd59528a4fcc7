"""Geometry helpers of the hot path. The reference's `dust3r/utils/geometry.py` offers general-purpose versions (`xy_grid` :15-37,
`geotrf` :40-101, `inv` :104-111); this package needs three call shapes only -- a pixel grid, "apply a batch of 4x4 (or one 4x4 / 3x3) to
points", and a matrix inverse -- plus `find_reciprocal_matches` (:345-361) on the GPU."""
import numpy as np
import torch


def xy_grid(W, H, device=None, origin=(0, 0), **arange_kw):
    """(H, W, 2) pixel grid, out[v, u] = (u + origin[0], v + origin[1]); a numpy array when device is None, else a tensor there."""
    if device is None:
        u, v = np.meshgrid(np.arange(origin[0], origin[0] + W, **arange_kw), np.arange(origin[1], origin[1] + H, **arange_kw), indexing='xy')
        return np.stack((u, v), axis=-1)
    u, v = torch.meshgrid(torch.arange(origin[0], origin[0] + W, device=device, **arange_kw),
                          torch.arange(origin[1], origin[1] + H, device=device, **arange_kw), indexing='xy')
    return torch.stack((u, v), dim=-1)


def geotrf(Trf, pts, ncol=None, norm=False):
    """Points (..., d) through transforms of size (d+1)x(d+1) (affine part applied, homogeneous row ignored) or d x d.
    Trf is one matrix, or a batch (B, ., .) matching the leading dimension of pts (B, ..., d). `norm` divides by the last
    coordinate (projection) and scales by it; `ncol` keeps the first columns. numpy in -> numpy out, torch in -> torch out."""
    is_np = isinstance(Trf, np.ndarray)
    pts = np.asarray(pts) if is_np else torch.as_tensor(pts, dtype=Trf.dtype, device=Trf.device)
    d = pts.shape[-1]
    lin = Trf[..., :d, :d]
    shift = Trf[..., :d, d] if Trf.shape[-1] == d + 1 else None
    if Trf.ndim == 3:                                  # one transform per leading index of pts
        flat = pts.reshape(pts.shape[0], -1, d)
        out = flat @ (lin.swapaxes(-1, -2))
        if shift is not None:
            out = out + shift[:, None, :]
    else:
        out = pts.reshape(-1, d) @ (lin.T if is_np else lin.transpose(-1, -2))
        if shift is not None:
            out = out + shift
    out = out.reshape(pts.shape)
    if norm:
        out = out / out[..., -1:]
        if norm != 1:
            out = out * norm
    return out[..., :ncol] if ncol else out


def inv(mat):
    if isinstance(mat, torch.Tensor):
        return torch.linalg.inv(mat)
    if isinstance(mat, np.ndarray):
        return np.linalg.inv(mat)
    raise ValueError(f'bad matrix type = {type(mat)}')


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
