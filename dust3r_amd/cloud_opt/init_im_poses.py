"""Entry points under the reference's names (`dust3r/cloud_opt/init_im_poses.py`: `init_minimum_spanning_tree`,
`init_from_known_poses`), implemented by the GPU scene bootstrap (cloud_opt/bootstrap.py + csrc/bootstrap.hip).
`compute_global_alignment(init='mst' | 'msp' | 'known_poses')` dispatches here (base_opt.py:275-287 of the reference)."""
from .bootstrap import bootstrap_from_known_poses, bootstrap_from_spanning_tree


def init_minimum_spanning_tree(scene, niter_PnP=10, **unused):
    return bootstrap_from_spanning_tree(scene, niter_PnP=niter_PnP)


def init_from_known_poses(scene, niter_PnP=10, min_conf_thr=3):
    return bootstrap_from_known_poses(scene, niter_PnP=niter_PnP, min_conf_thr=min_conf_thr)
