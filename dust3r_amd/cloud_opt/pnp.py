"""Dependency-free PnP-RANSAC (numpy). Stands in for `cv2.solvePnPRansac(..., flags=SOLVEPNP_SQPNP)`
+ `cv2.Rodrigues`, which the reference calls at dust3r/cloud_opt/init_im_poses.py:272-285 and
pair_viewer.py:55-60 (OpenCV is not a dependency of this framework).

Model: world->camera (R, T) with pixel = K (R X + T). Hypotheses come from a 6-point DLT on
normalised image coordinates projected onto SO(3); the best consensus set is refined by a few
Gauss-Newton steps on the reprojection error. OpenCV's RANSAC is RNG dependent, so this boundary has
no bit-level parity by construction; both return an inlier-consensus pose on low-outlier pointmaps.
"""
import numpy as np


def rodrigues_to_rotmat(rvec):
    rvec = np.asarray(rvec, np.float64).ravel()
    th = np.linalg.norm(rvec)
    if th < 1e-12:
        return np.eye(3)
    k = rvec / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * (K @ K)


def rotmat_to_rodrigues(R):
    R = np.asarray(R, np.float64)
    c = np.clip((np.trace(R) - 1) / 2, -1, 1)
    th = np.arccos(c)
    if th < 1e-12:
        return np.zeros(3)
    if np.pi - th < 1e-6:
        # near pi: axis from the dominant column of (R + I)
        A = (R + np.eye(3)) / 2
        k = np.sqrt(np.clip(np.diag(A), 0, None))
        i = int(np.argmax(k))
        k = A[:, i] / max(k[i], 1e-12)
        return th * k / np.linalg.norm(k)
    w = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]]) / (2 * np.sin(th))
    return th * w


def pose_from_dlt_normal(AtA, mean_point=None):
    """12x12 normal matrix of the DLT system (unknown = the rows of [R | T] stacked) -> (R, T) or None. The null vector is the
    eigenvector of the smallest eigenvalue, its 3x3 part is projected onto SO(3). `mean_point`: centroid of the world points,
    used for the cheirality sign when the points themselves are not at hand (the GPU path only has the moments)."""
    if not np.isfinite(AtA).all():
        return None
    try:
        _, V = np.linalg.eigh(AtA)
        P = V[:, 0].reshape(3, 4)
        U, S, Vt2 = np.linalg.svd(P[:, :3])
    except np.linalg.LinAlgError:
        return None
    if S.mean() < 1e-12:
        return None
    R = U @ Vt2
    sgn = 1.0
    if np.linalg.det(R) < 0:
        R, sgn = -R, -1.0
    T = sgn * P[:, 3] / S.mean()
    if mean_point is not None and (R[2] @ mean_point + T[2]) < 0:
        return None
    return R, T


def _dlt_pose(X, xn):
    """X (n,3) world points, xn (n,2) normalised image coords -> (R, T) or None."""
    n = len(X)
    Xh = np.concatenate((X, np.ones((n, 1))), axis=1)
    A = np.zeros((2 * n, 12))
    A[0::2, 0:4] = -Xh
    A[0::2, 8:12] = xn[:, 0:1] * Xh
    A[1::2, 4:8] = -Xh
    A[1::2, 8:12] = xn[:, 1:2] * Xh
    # through the 12x12 normal matrix (a full SVD of A builds the 2n x 2n left factor: 43 s per call on a 20 000-point consensus set)
    sol = pose_from_dlt_normal(A.T @ A)
    if sol is None:
        return None
    R, T = sol
    if np.median((X @ R.T + T)[:, 2]) < 0:      # cheirality: most points in front of the camera
        return None
    return R, T


def dlt_pose_batch(X, xn, sel):
    """`_dlt_pose` for a stack of point sets in one pass of batched LAPACK calls: X (b, p, 3) world points, xn (b, p, 2) normalised image
    coordinates, sel (b, p) bool = the points of set b that take part (unselected rows enter the normal matrix with weight zero, which is the
    system without them). Returns R (b, 3, 3), T (b, 3), valid (b,): the same null vector / SO(3) projection / cheirality rule per set.
    The 100-view spanning-tree initialisation forms 960 six-to-twelve-point hypotheses: one call instead of 960 (0.06 s of numpy call overhead)."""
    X = np.asarray(X, np.float64)
    xn = np.asarray(xn, np.float64)
    b, p = X.shape[:2]
    selb = np.asarray(sel, bool)
    w = selb.astype(np.float64)
    # unselected rows are ZEROED, not multiplied by a zero weight: NaN * 0 and inf * 0 are NaN, and a non-finite low-confidence candidate that the scalar
    # path never reads must not reject the hypothesis (bootstrap.solve_pnp_batch passes whole candidate windows)
    X = np.where(selb[:, :, None], X, 0.0)
    xn = np.where(selb[:, :, None], xn, 0.0)
    Xh = np.concatenate((X, np.ones((b, p, 1))), axis=2) * w[:, :, None]
    a1 = np.zeros((b, p, 12))
    a2 = np.zeros((b, p, 12))
    a1[:, :, 0:4] = -Xh
    a1[:, :, 8:12] = xn[:, :, 0:1] * Xh
    a2[:, :, 4:8] = -Xh
    a2[:, :, 8:12] = xn[:, :, 1:2] * Xh
    AtA = np.einsum('bpi,bpj->bij', a1, a1) + np.einsum('bpi,bpj->bij', a2, a2)
    valid = np.isfinite(AtA).all(axis=(1, 2)) & (w.sum(axis=1) >= 6)
    AtA[~valid] = np.eye(12)
    _, V = np.linalg.eigh(AtA)
    P = V[:, :, 0].reshape(b, 3, 4)
    U, S, Vt = np.linalg.svd(P[:, :, :3])
    sm = S.mean(axis=1)
    valid &= sm >= 1e-12
    sm = np.where(sm < 1e-12, 1.0, sm)
    R = U @ Vt
    sgn = np.where(np.linalg.det(R) < 0, -1.0, 1.0)
    R = R * sgn[:, None, None]
    T = sgn[:, None] * P[:, :, 3] / sm[:, None]
    z = np.einsum('bpk,bk->bp', X, R[:, 2, :]) + T[:, 2:3]
    z = np.where(selb, z, np.nan)
    with np.errstate(all='ignore'):
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter('ignore', RuntimeWarning)
            med = np.nanmedian(z, axis=1)
    valid &= np.nan_to_num(med, nan=-1.0) >= 0          # cheirality: most points in front of the camera
    return R, T, valid


def _project(X, R, T, K):
    Xc = X @ R.T + T
    z = np.where(np.abs(Xc[:, 2]) < 1e-12, 1e-12, Xc[:, 2])
    return np.stack((K[0, 0] * Xc[:, 0] / z + K[0, 2], K[1, 1] * Xc[:, 1] / z + K[1, 2]), axis=1), Xc


def _refine(X, pix, K, R, T, iters=10):
    """Gauss-Newton on reprojection error over (rotation vector increment, translation)."""
    for _ in range(iters):
        proj, Xc = _project(X, R, T, K)
        r = (proj - pix).reshape(-1)
        x, y, z = Xc[:, 0], Xc[:, 1], np.where(np.abs(Xc[:, 2]) < 1e-12, 1e-12, Xc[:, 2])
        fx, fy = K[0, 0], K[1, 1]
        # d proj / d Xc
        J_pc = np.zeros((len(X), 2, 3))
        J_pc[:, 0, 0], J_pc[:, 0, 2] = fx / z, -fx * x / z ** 2
        J_pc[:, 1, 1], J_pc[:, 1, 2] = fy / z, -fy * y / z ** 2
        # d Xc / d (w, t) with R <- exp([w]x) R :  dXc = -[Xc - T]x w + t
        Xr = Xc - T
        skew = np.zeros((len(X), 3, 3))
        skew[:, 0, 1], skew[:, 0, 2] = Xr[:, 2], -Xr[:, 1]
        skew[:, 1, 0], skew[:, 1, 2] = -Xr[:, 2], Xr[:, 0]
        skew[:, 2, 0], skew[:, 2, 1] = Xr[:, 1], -Xr[:, 0]
        J = np.concatenate((J_pc @ skew, J_pc), axis=2).reshape(-1, 6)
        H = J.T @ J + 1e-9 * np.eye(6)
        try:
            d = np.linalg.solve(H, -J.T @ r)
        except np.linalg.LinAlgError:
            break
        R = rodrigues_to_rotmat(d[:3]) @ R
        T = T + d[3:]
        if np.linalg.norm(d) < 1e-10:
            break
    return R, T


def solve_pnp_ransac(pts3d, pixels, K, iterations=100, reproj_err=5.0, seed=0, min_sample=6):
    """Returns (success, R (3,3) world->cam, T (3,), inlier indices)."""
    X = np.asarray(pts3d, np.float64).reshape(-1, 3)
    pix = np.asarray(pixels, np.float64).reshape(-1, 2)
    K = np.asarray(K, np.float64)
    n = len(X)
    if n < min_sample:
        return False, None, None, None
    xn = np.stack(((pix[:, 0] - K[0, 2]) / K[0, 0], (pix[:, 1] - K[1, 2]) / K[1, 1]), axis=1)
    rng = np.random.RandomState(seed)
    best = (0, None, None, None)
    for _ in range(max(1, int(iterations))):
        idx = rng.choice(n, min_sample, replace=False)
        hyp = _dlt_pose(X[idx], xn[idx])
        if hyp is None:
            continue
        proj, Xc = _project(X, hyp[0], hyp[1], K)
        inl = np.nonzero((np.linalg.norm(proj - pix, axis=1) < reproj_err) & (Xc[:, 2] > 0))[0]
        if len(inl) > best[0]:
            best = (len(inl), hyp[0], hyp[1], inl)
    if best[0] < min_sample:
        return False, None, None, None
    _, R, T, inl = best
    for _ in range(2):   # refit on the consensus set, then polish
        sub = inl if len(inl) <= 20000 else inl[np.linspace(0, len(inl) - 1, 20000).astype(int)]
        hyp = _dlt_pose(X[sub], xn[sub])
        if hyp is not None:
            R2, T2 = _refine(X[sub], pix[sub], K, hyp[0], hyp[1])
            proj, Xc = _project(X, R2, T2, K)
            inl2 = np.nonzero((np.linalg.norm(proj - pix, axis=1) < reproj_err) & (Xc[:, 2] > 0))[0]
            if len(inl2) >= len(inl):
                R, T, inl = R2, T2, inl2
    return True, R, T, inl
