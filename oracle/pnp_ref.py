"""TEST INFRASTRUCTURE (oracle) -- an independent stand-in for `cv2.solvePnPRansac(objectPoints, imagePoints, K, None,
iterationsCount=..., reprojectionError=..., flags=cv2.SOLVEPNP_SQPNP)` + `cv2.Rodrigues`, which the reference calls at
/root/reference/dust3r/cloud_opt/init_im_poses.py:247-287 (`fast_pnp`) and /root/reference/dust3r/cloud_opt/pair_viewer.py:55-63.

OpenCV (opencv-python 4.x, un-vendored dependency of the reference) is absent from this image, so its algorithm is restated from its
published description, NOT from the product's solver (nothing under oracle/ imports the product's algorithms -- only its seeded input generators, dust3r_amd.synthetic; tests/test_host_cpu.py enforces it):

  * RANSAC as `cv::solvePnPRansac` runs it: minimal samples drawn with a seeded generator, a model per sample, inliers = points whose
    reprojection error is below `reprojectionError` pixels, `iterationsCount` iterations at most with the standard adaptive stopping rule
    at confidence 0.99 (`RANSACUpdateNumIters`), the best model = the largest consensus set;
  * the minimal model: classical camera RESECTION -- the 2n x 12 direct-linear-transform system in PIXEL coordinates with Hartley
    normalisation of both point sets, P = K [R | t] recovered from the null vector, K^-1 P projected onto SO(3) (the product solves a 12 x 12
    normal-equation eigenproblem on K-normalised rays instead);
  * the final model: Levenberg-Marquardt on the reprojection error of the consensus set over (Rodrigues vector, t) with
    `scipy.optimize.least_squares` -- what OpenCV does after RANSAC (`solvePnP(..., SOLVEPNP_ITERATIVE)`-style refinement of the inlier set;
    SQPnP itself returns the global minimiser of the same algebraic cost, which LM from the RANSAC model reaches on low-outlier pointmaps).

Parity status: UNPINNED against OpenCV itself (absent; RNG-dependent); what this oracle pins is that the product's pose agrees with an
independently written consensus + least-squares solver on the same points (tests/test_aligner_gpu.py, tests/test_bootstrap_cpu.py).
Model: world -> camera (R, t), pixel ~ K (R X + t)."""
import numpy as np
from scipy.optimize import least_squares


def rodrigues(rvec):
    """cv2.Rodrigues(rvec)[0]: axis-angle -> rotation matrix."""
    r = np.asarray(rvec, np.float64).reshape(3)
    th = float(np.linalg.norm(r))
    if th < 1e-14:
        return np.eye(3)
    k = r / th
    Kx = np.array([[0.0, -k[2], k[1]], [k[2], 0.0, -k[0]], [-k[1], k[0], 0.0]])
    return np.cos(th) * np.eye(3) + (1.0 - np.cos(th)) * np.outer(k, k) + np.sin(th) * Kx


def rodrigues_inv(R):
    """Rotation matrix -> axis-angle through the quaternion (robust near 0 and pi)."""
    R = np.asarray(R, np.float64)
    q = np.empty(4)
    tr = np.trace(R)
    if tr > 0:
        s = np.sqrt(tr + 1.0) * 2
        q[:] = (0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s)
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0) * 2
        q[0] = (R[k, j] - R[j, k]) / s
        q[1 + i] = 0.25 * s
        q[1 + j] = (R[j, i] + R[i, j]) / s
        q[1 + k] = (R[k, i] + R[i, k]) / s
    if q[0] < 0:
        q = -q
    n = float(np.linalg.norm(q[1:]))
    if n < 1e-14:
        return np.zeros(3)
    return q[1:] / n * (2.0 * np.arctan2(n, q[0]))


def _hartley(x):
    c = x.mean(axis=0)
    d = np.sqrt(((x - c) ** 2).sum(axis=1)).mean()
    s = np.sqrt(x.shape[1]) / max(d, 1e-12)
    T = np.eye(x.shape[1] + 1)
    T[:-1, :-1] *= s
    T[:-1, -1] = -s * c
    return (x - c) * s, T


def resection(X, uv, K):
    """Camera resection from n >= 6 correspondences: (R, t) or None."""
    Xn, TX = _hartley(X)
    un, Tu = _hartley(uv)
    n = len(X)
    A = np.zeros((2 * n, 12))
    Xh = np.concatenate((Xn, np.ones((n, 1))), axis=1)
    A[0::2, 0:4] = Xh
    A[0::2, 8:12] = -un[:, :1] * Xh
    A[1::2, 4:8] = Xh
    A[1::2, 8:12] = -un[:, 1:] * Xh
    try:
        _, _, Vt = np.linalg.svd(A, full_matrices=False)
    except np.linalg.LinAlgError:
        return None
    P = np.linalg.inv(Tu) @ Vt[-1].reshape(3, 4) @ TX               # pixel ~ P [X; 1]
    M = np.linalg.inv(K) @ P                                          # ~ s [R | t]
    try:
        U, S, Wt = np.linalg.svd(M[:, :3])
    except np.linalg.LinAlgError:
        return None
    if S.min() < 1e-12 * max(S.max(), 1e-300):
        return None
    R = U @ Wt
    s = S.mean()
    if np.linalg.det(R) < 0:
        R, s = -R, -s
    t = M[:, 3] / s
    if np.median(X @ R[2] + t[2]) <= 0:                               # points behind the camera: the other sign of the null vector
        return None
    return R, t


def reproject(X, R, t, K):
    c = X @ R.T + t
    z = np.where(np.abs(c[:, 2:3]) < 1e-12, 1e-12, c[:, 2:3])
    p = c[:, :2] / z
    return p * np.array([K[0, 0], K[1, 1]]) + np.array([K[0, 2], K[1, 2]]) + np.stack((K[0, 1] * p[:, 1], np.zeros(len(X))), axis=1)


def refine(X, uv, K, R, t):
    """Levenberg-Marquardt on the reprojection error over (Rodrigues vector, t)."""
    def res(pv):
        return (reproject(X, rodrigues(pv[:3]), pv[3:], K) - uv).ravel()
    sol = least_squares(res, np.concatenate((rodrigues_inv(R), t)), method='lm', xtol=1e-12, ftol=1e-12, gtol=1e-12, max_nfev=200)
    return rodrigues(sol.x[:3]), sol.x[3:]


def solve_pnp_ransac(objectPoints, imagePoints, cameraMatrix, iterationsCount=100, reprojectionError=8.0, confidence=0.99, sample=6, seed=0):
    """-> (ok, rvec (3, 1), tvec (3, 1), inliers (k, 1) int32), the return structure of cv2.solvePnPRansac."""
    X = np.asarray(objectPoints, np.float64).reshape(-1, 3)
    uv = np.asarray(imagePoints, np.float64).reshape(-1, 2)
    K = np.asarray(cameraMatrix, np.float64)
    n = len(X)
    if n < sample:
        return False, None, None, None
    rng = np.random.RandomState(seed)
    best, best_cnt, iters, it = None, 0, int(iterationsCount), 0
    while it < iters:
        it += 1
        idx = rng.choice(n, sample, replace=False)
        model = resection(X[idx], uv[idx], K)
        if model is None:
            continue
        err = np.linalg.norm(reproject(X, model[0], model[1], K) - uv, axis=1)
        inl = err < reprojectionError
        cnt = int(inl.sum())
        if cnt > best_cnt:
            best, best_cnt = inl, cnt
            w = cnt / n                                               # cv::RANSACUpdateNumIters
            denom = np.log(max(1.0 - w ** sample, 1e-300))
            if denom < 0:
                iters = min(iters, int(np.ceil(np.log(1.0 - confidence) / denom)) if w < 1.0 else it)
    if best is None or best_cnt < sample:
        return False, None, None, None
    R, t = resection(X[best], uv[best], K) or (None, None)
    if R is None:
        return False, None, None, None
    R, t = refine(X[best], uv[best], K, R, t)
    # the consensus set re-selected around the refined model and the model refined on it, until the set no longer changes (a handful of rounds): the
    # least-squares pose whose inlier set is its own -- independent of which minimal sample RANSAC happened to start from
    for _ in range(10):
        inl = np.linalg.norm(reproject(X, R, t, K) - uv, axis=1) < reprojectionError
        if int(inl.sum()) < sample or np.array_equal(inl, best):
            break
        R, t = refine(X[inl], uv[inl], K, R, t)
        best = inl
    return True, rodrigues_inv(R).reshape(3, 1), np.asarray(t, np.float64).reshape(3, 1), np.nonzero(best)[0].astype(np.int32).reshape(-1, 1)
