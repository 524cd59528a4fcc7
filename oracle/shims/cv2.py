"""Stub of the few cv2 entry points on the reference's hot path
(`dust3r/cloud_opt/init_im_poses.py:272-285`, `pair_viewer.py:55-60`).
solvePnPRansac/Rodrigues delegate to the framework's own dependency-free PnP
(dust3r_amd/cloud_opt/pnp.py): OpenCV's SQPnP-RANSAC is RNG dependent and absent
here, so this boundary is PARITY UNPINNED by construction."""
import numpy as np

IMREAD_COLOR = 1
IMREAD_ANYDEPTH = 2
COLOR_BGR2RGB = 4
SOLVEPNP_SQPNP = 8


def solvePnPRansac(objectPoints, imagePoints, cameraMatrix, distCoeffs, iterationsCount=100,
                   reprojectionError=8.0, flags=0, **kw):
    from dust3r_amd.cloud_opt.pnp import solve_pnp_ransac
    ok, R, T, inl = solve_pnp_ransac(np.asarray(objectPoints, np.float64), np.asarray(imagePoints, np.float64),
                                     np.asarray(cameraMatrix, np.float64), iterations=iterationsCount,
                                     reproj_err=reprojectionError)
    if not ok:
        return False, None, None, None
    from dust3r_amd.cloud_opt.pnp import rotmat_to_rodrigues
    return True, rotmat_to_rodrigues(R).reshape(3, 1), T.reshape(3, 1), inl.reshape(-1, 1)


def Rodrigues(rvec):
    from dust3r_amd.cloud_opt.pnp import rodrigues_to_rotmat
    return rodrigues_to_rotmat(np.asarray(rvec, np.float64).ravel()), None


def imread(*a, **k):
    raise NotImplementedError('cv2 stub')


def cvtColor(*a, **k):
    raise NotImplementedError('cv2 stub')
