"""ORACLE (test infrastructure only) -- pure-torch restatement of the subset of
`roma` (PyPI, unpinned in /root/reference/requirements.txt:3) that the
reference's hot path calls:

  * `roma.RigidUnitQuat(Q, T).normalize().to_homogeneous()`   dust3r/cloud_opt/base_opt.py:154
  * `roma.rotmat_to_unitquat(R)`                              dust3r/cloud_opt/base_opt.py:169
  * `roma.rigid_points_registration(x, y, weights, compute_scaling=True)`
                                 dust3r/cloud_opt/init_im_poses.py:221-222,315

PARITY UNPINNED: roma is a third-party dependency that is not installed here
and cannot be fetched; the published algorithms are restated.
Quaternion order is XYZW (scalar last).
"""
import torch


def unitquat_to_rotmat(quat):
    x, y, z, w = torch.unbind(quat, dim=-1)
    xx, yy, zz = x * x, y * y, z * z
    xy, xz, yz = x * y, x * z, y * z
    xw, yw, zw = x * w, y * w, z * w
    rows = [
        torch.stack((1 - 2 * (yy + zz), 2 * (xy - zw), 2 * (xz + yw)), dim=-1),
        torch.stack((2 * (xy + zw), 1 - 2 * (xx + zz), 2 * (yz - xw)), dim=-1),
        torch.stack((2 * (xz - yw), 2 * (yz + xw), 1 - 2 * (xx + yy)), dim=-1),
    ]
    return torch.stack(rows, dim=-2)


def rotmat_to_unitquat(R):
    """Batched Shepperd-style extraction; returns XYZW with the largest-magnitude
    component positive (any sign is equivalent: R(q) == R(-q))."""
    R = torch.as_tensor(R)
    batch = R.shape[:-2]
    m = R.reshape(-1, 3, 3)
    m00, m11, m22 = m[:, 0, 0], m[:, 1, 1], m[:, 2, 2]
    tr = m00 + m11 + m22
    # four candidate "4 q_k^2" values: x, y, z, w
    dec = torch.stack((1 + m00 - m11 - m22, 1 - m00 + m11 - m22, 1 - m00 - m11 + m22, 1 + tr), dim=-1)
    choice = dec.argmax(dim=-1)
    q = m.new_zeros((m.shape[0], 4))
    for n in range(m.shape[0]):
        c = int(choice[n])
        M = m[n]
        if c == 3:
            q[n] = torch.stack((M[2, 1] - M[1, 2], M[0, 2] - M[2, 0], M[1, 0] - M[0, 1], dec[n, 3]))
        elif c == 0:
            q[n] = torch.stack((dec[n, 0], M[1, 0] + M[0, 1], M[0, 2] + M[2, 0], M[2, 1] - M[1, 2]))
        elif c == 1:
            q[n] = torch.stack((M[1, 0] + M[0, 1], dec[n, 1], M[2, 1] + M[1, 2], M[0, 2] - M[2, 0]))
        else:
            q[n] = torch.stack((M[0, 2] + M[2, 0], M[2, 1] + M[1, 2], dec[n, 2], M[1, 0] - M[0, 1]))
    q = q / q.norm(dim=-1, keepdim=True)
    return q.reshape(*batch, 4)


class RigidUnitQuat:
    """Rigid transform parameterised by (unit quaternion XYZW, translation)."""

    def __init__(self, linear, translation):
        self.linear = linear
        self.translation = translation

    def normalize(self):
        return RigidUnitQuat(self.linear / torch.norm(self.linear, dim=-1, keepdim=True), self.translation)

    def to_homogeneous(self):
        R = unitquat_to_rotmat(self.linear)
        batch = R.shape[:-2]
        H = torch.zeros(batch + (4, 4), dtype=R.dtype, device=R.device)
        H[..., :3, :3] = R
        H[..., :3, 3] = self.translation
        H[..., 3, 3] = 1
        return H


def special_procrustes(M, return_singular_values=False):
    """argmin_{R in SO(3)} ||R - M||_F via SVD with determinant correction."""
    U, S, Vh = torch.linalg.svd(M)
    det = torch.det(U @ Vh)
    D = torch.ones_like(S)
    D[..., -1] = det
    R = (U * D[..., None, :]) @ Vh
    if return_singular_values:
        return R, S * D
    return R


def rigid_points_registration(x, y, weights=None, compute_scaling=False):
    """(R, t[, s]) minimising sum_k w_k || s R x_k + t - y_k ||^2 (weighted Umeyama)."""
    if weights is None:
        xmean = x.mean(dim=-2, keepdim=True)
        ymean = y.mean(dim=-2, keepdim=True)
        xhat, yhat = x - xmean, y - ymean
        M = yhat.transpose(-1, -2) @ xhat
        xnorm2 = xhat.square().sum(dim=(-1, -2))
    else:
        w = weights[..., None]
        n = weights.sum(dim=-1)[..., None, None]
        xmean = (w * x).sum(dim=-2, keepdim=True) / n
        ymean = (w * y).sum(dim=-2, keepdim=True) / n
        xhat, yhat = x - xmean, y - ymean
        M = yhat.transpose(-1, -2) @ (w * xhat)
        xnorm2 = (w * xhat.square()).sum(dim=(-1, -2))
    if compute_scaling:
        R, DS = special_procrustes(M, return_singular_values=True)
        scale = DS.sum(dim=-1) / xnorm2
        t = ymean.squeeze(-2) - scale[..., None] * (R @ xmean.transpose(-1, -2)).squeeze(-1)
        return R, t, scale
    R = special_procrustes(M)
    t = ymean.squeeze(-2) - (R @ xmean.transpose(-1, -2)).squeeze(-1)
    return R, t
