"""Rigid-motion helpers used by the aligner's host side: quaternions (XYZW, scalar last) and
weighted similarity Procrustes. These provide what the reference takes from the third-party
`roma` package (`roma.rotmat_to_unitquat`, `roma.unitquat_to_rotmat`,
`roma.rigid_points_registration(..., compute_scaling=True)`; call sites
dust3r/cloud_opt/base_opt.py:154,169 and init_im_poses.py:221-222,315)."""
import torch


def unitquat_to_rotmat(q):
    x, y, z, w = torch.unbind(q, dim=-1)
    return torch.stack((
        torch.stack((1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)), dim=-1),
        torch.stack((2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)), dim=-1),
        torch.stack((2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)), dim=-1)), dim=-2)


def rotmat_to_unitquat(R):
    """(..., 3, 3) -> (..., 4) XYZW, numerically stable largest-component branch."""
    R = torch.as_tensor(R)
    m = R.reshape(-1, 3, 3)
    d0, d1, d2 = m[:, 0, 0], m[:, 1, 1], m[:, 2, 2]
    cand = torch.stack((1 + d0 - d1 - d2, 1 - d0 + d1 - d2, 1 - d0 - d1 + d2, 1 + d0 + d1 + d2), dim=-1)
    qs = torch.stack((
        torch.stack((cand[:, 0], m[:, 1, 0] + m[:, 0, 1], m[:, 0, 2] + m[:, 2, 0], m[:, 2, 1] - m[:, 1, 2]), dim=-1),
        torch.stack((m[:, 1, 0] + m[:, 0, 1], cand[:, 1], m[:, 2, 1] + m[:, 1, 2], m[:, 0, 2] - m[:, 2, 0]), dim=-1),
        torch.stack((m[:, 0, 2] + m[:, 2, 0], m[:, 2, 1] + m[:, 1, 2], cand[:, 2], m[:, 1, 0] - m[:, 0, 1]), dim=-1),
        torch.stack((m[:, 2, 1] - m[:, 1, 2], m[:, 0, 2] - m[:, 2, 0], m[:, 1, 0] - m[:, 0, 1], cand[:, 3]), dim=-1)), dim=1)
    pick = cand.argmax(dim=-1)
    q = qs[torch.arange(m.shape[0]), pick]
    q = q / q.norm(dim=-1, keepdim=True)
    return q.reshape(*R.shape[:-2], 4)


def quat_translation_to_homogeneous(Q, T):
    """Normalises Q; returns (..., 4, 4) [R T; 0 1]."""
    R = unitquat_to_rotmat(Q / Q.norm(dim=-1, keepdim=True))
    H = torch.zeros(R.shape[:-2] + (4, 4), dtype=R.dtype, device=R.device)
    H[..., :3, :3] = R
    H[..., :3, 3] = T
    H[..., 3, 3] = 1
    return H


def rigid_points_registration(x, y, weights=None, compute_scaling=False):
    """argmin sum_k w_k |s R x_k + t - y_k|^2 (weighted Umeyama). Returns (R, t[, s])."""
    if weights is None:
        weights = torch.ones(x.shape[:-1], dtype=x.dtype, device=x.device)
    w = weights[..., None]
    n = weights.sum(dim=-1)[..., None, None]
    xm = (w * x).sum(dim=-2, keepdim=True) / n
    ym = (w * y).sum(dim=-2, keepdim=True) / n
    xh, yh = x - xm, y - ym
    M = yh.transpose(-1, -2) @ (w * xh)
    U, S, Vh = torch.linalg.svd(M)
    D = torch.ones_like(S)
    D[..., -1] = torch.det(U @ Vh)
    R = (U * D[..., None, :]) @ Vh
    if compute_scaling:
        s = (S * D).sum(dim=-1) / (w * xh.square()).sum(dim=(-1, -2))
        t = ym.squeeze(-2) - s[..., None] * (R @ xm.transpose(-1, -2)).squeeze(-1)
        return R, t, s
    t = ym.squeeze(-2) - (R @ xm.transpose(-1, -2)).squeeze(-1)
    return R, t
