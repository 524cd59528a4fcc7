"""TEST INFRASTRUCTURE (oracle) -- stand-in for the few cv2 entry points on the reference's hot path
(/root/reference/dust3r/cloud_opt/init_im_poses.py:272-285, pair_viewer.py:55-60), so that the UNMODIFIED reference files can be
imported and run here (oracle/ref_import.py). OpenCV is absent from this image. solvePnPRansac / Rodrigues go to oracle/pnp_ref.py, an
independent restatement of the published algorithm (seeded RANSAC with cv2's parameters + resection + Levenberg-Marquardt refinement);
nothing under oracle/ imports the product's algorithms (only dust3r_amd.synthetic, the seeded input generators): goldens generated through this
shim are independent evidence for its PnP."""
import numpy as np

from oracle import pnp_ref

IMREAD_COLOR = 1
IMREAD_ANYDEPTH = 2
COLOR_BGR2RGB = 4
SOLVEPNP_SQPNP = 8


def solvePnPRansac(objectPoints, imagePoints, cameraMatrix, distCoeffs, iterationsCount=100,
                   reprojectionError=8.0, flags=0, **kw):
    return pnp_ref.solve_pnp_ransac(objectPoints, imagePoints, cameraMatrix, iterationsCount=iterationsCount, reprojectionError=reprojectionError)


def Rodrigues(rvec):
    return pnp_ref.rodrigues(np.asarray(rvec, np.float64).ravel()), None


def imread(*a, **k):
    raise NotImplementedError('cv2 stub')


def cvtColor(*a, **k):
    raise NotImplementedError('cv2 stub')
