"""One-shot initialisation of the global aligner -- host-side mirror of the reference
`dust3r/cloud_opt/init_im_poses.py` (`init_minimum_spanning_tree`, `init_from_known_poses`,
`init_from_pts3d`, `minimum_spanning_tree`, `fast_pnp`, `align_multiple_poses`, ...).

Same procedure: maximum-confidence spanning tree over the pair graph (SciPy MST on negated edge
scores), chained weighted similarity Procrustes along the tree, Weiszfeld focal estimation, PnP for
views that never acted as the first image of a tree edge, then per-edge Procrustes to initialise
the pairwise poses and the scale normalisation. roma / cv2 calls are replaced by utils/rigid.py and
cloud_opt/pnp.py (both dependency-free). Runs with torch ops on whatever device the scene lives on.
"""
from functools import lru_cache

import numpy as np
import scipy.sparse as sp
import torch
from tqdm import tqdm

from ..post_process import estimate_focal_knowing_depth
from ..utils.device import to_numpy
from ..utils.geometry import geotrf, get_med_dist_between_poses, inv
from ..utils.rigid import rigid_points_registration as _umeyama
from .commons import compute_edge_scores, edge_str, i_j_ij
from .pnp import solve_pnp_ransac


def rigid_points_registration(pts1, pts2, conf):
    R, T, s = _umeyama(pts1.reshape(-1, 3), pts2.reshape(-1, 3), weights=conf.ravel(), compute_scaling=True)
    return s, R, T


def sRT_to_4x4(scale, R, T, device):
    trf = torch.eye(4, device=device)
    trf[:3, :3] = R * scale
    trf[:3, 3] = T.ravel()
    return trf


def estimate_focal(pts3d_i, pp=None):
    if pp is None:
        H, W, THREE = pts3d_i.shape
        assert THREE == 3
        pp = torch.tensor((W / 2, H / 2), device=pts3d_i.device)
    return float(estimate_focal_knowing_depth(pts3d_i.unsqueeze(0), pp.unsqueeze(0), focal_mode='weiszfeld').ravel())


@lru_cache(maxsize=None)
def pixel_grid(H, W):
    return np.mgrid[:W, :H].T.astype(np.float32)


def fast_pnp(pts3d, focal, msk, device, pp=None, niter_PnP=10):
    """Camera pose (cam-to-world) and focal of one view from its world-frame pointmap."""
    if msk.sum() < 4:
        return None
    pts3d, msk = map(to_numpy, (pts3d, msk))
    H, W, THREE = pts3d.shape
    assert THREE == 3
    pixels = pixel_grid(H, W)
    if focal is None:
        S = max(W, H)
        tentative_focals = np.geomspace(S / 2, S * 3, 21)
    else:
        tentative_focals = [focal]
    pp = (W / 2, H / 2) if pp is None else to_numpy(pp)
    best = (0,)
    for f in tentative_focals:
        K = np.float32([(f, 0, pp[0]), (0, f, pp[1]), (0, 0, 1)])
        success, R, T, inliers = solve_pnp_ransac(pts3d[msk], pixels[msk], K, iterations=niter_PnP, reproj_err=5)
        if success and len(inliers) > best[0]:
            best = (len(inliers), R, T, f)
    if not best[0]:
        return None
    _, R, T, best_focal = best
    R, T = torch.from_numpy(R).float(), torch.from_numpy(T).float()
    return best_focal, inv(sRT_to_4x4(1, R, T, device))


def get_known_poses(self):
    if self.has_im_poses:
        msk = torch.tensor([not self.im_poses.requires_grad] * self.n_imgs)
        return msk.sum(), msk, self.get_im_poses()
    return 0, None, None


def get_known_focals(self):
    if self.has_im_poses:
        msk = self.get_known_focal_mask()
        return msk.sum(), msk, self.get_focals()
    return 0, None, None


def align_multiple_poses(src_poses, target_poses):
    N = len(src_poses)
    assert src_poses.shape == target_poses.shape == (N, 4, 4)

    def center_and_z(poses):
        eps = get_med_dist_between_poses(poses) / 100
        return torch.cat((poses[:, :3, 3], poses[:, :3, 3] + eps * poses[:, :3, 2]))
    R, T, s = _umeyama(center_and_z(src_poses), center_and_z(target_poses), compute_scaling=True)
    return s, R, T


def dict_to_sparse_graph(dic):
    n_imgs = max(max(e) for e in dic) + 1
    res = sp.dok_array((n_imgs, n_imgs))
    for edge, value in dic.items():
        res[edge] = value
    return res


@torch.no_grad()
def init_from_known_poses(self, niter_PnP=10, min_conf_thr=3):
    device = self.device
    nkp, known_poses_msk, known_poses = get_known_poses(self)
    assert nkp == self.n_imgs, 'not all poses are known'
    nkf, _, im_focals = get_known_focals(self)
    assert nkf == self.n_imgs
    im_pp = self.get_principal_points()
    best_depthmaps = {}
    for e, (i, j) in enumerate(tqdm(self.edges, disable=not self.verbose)):
        i_j = edge_str(i, j)
        P1 = torch.eye(4, device=device)
        msk = self.conf_i[i_j] > min(min_conf_thr, self.conf_i[i_j].min() - 0.1)
        _, P2 = fast_pnp(self.pred_j[i_j], float(im_focals[i].mean()), pp=im_pp[i], msk=msk, device=device, niter_PnP=niter_PnP)
        s, R, T = align_multiple_poses(torch.stack((P1, P2.to(device))), known_poses[[i, j]])
        self._set_pose(self.pw_poses, e, R, T, scale=s)
        score = float(self.conf_i[i_j].mean())
        if score > best_depthmaps.get(i, (0,))[0]:
            best_depthmaps[i] = score, i_j, s
    for n in range(self.n_imgs):
        assert known_poses_msk[n]
        _, i_j, scale = best_depthmaps[n]
        self._set_depthmap(n, self.pred_i[i_j][:, :, 2] * scale)


@torch.no_grad()
def init_minimum_spanning_tree(self, **kw):
    pts3d, _, im_focals, im_poses = minimum_spanning_tree(self.imshapes, self.edges, self.pred_i, self.pred_j, self.conf_i, self.conf_j,
                                                          self.im_conf, self.min_conf_thr, self.device,
                                                          has_im_poses=self.has_im_poses, verbose=self.verbose, **kw)
    return init_from_pts3d(self, pts3d, im_focals, im_poses)


def init_from_pts3d(self, pts3d, im_focals, im_poses):
    nkp, known_poses_msk, known_poses = get_known_poses(self)
    if nkp == 1:
        raise NotImplementedError('Would be simpler to just align everything afterwards on the single known pose')
    elif nkp > 1:
        s, R, T = align_multiple_poses(im_poses[known_poses_msk], known_poses[known_poses_msk])
        trf = sRT_to_4x4(s, R, T, device=known_poses.device)
        im_poses = trf @ im_poses
        im_poses[:, :3, :3] /= s
        for img_pts3d in pts3d:
            img_pts3d[:] = geotrf(trf, img_pts3d)
    for e, (i, j) in enumerate(self.edges):
        i_j = edge_str(i, j)
        s, R, T = rigid_points_registration(self.pred_i[i_j], pts3d[i], conf=self.conf_i[i_j])
        self._set_pose(self.pw_poses, e, R, T, scale=s)
    s_factor = self.get_pw_norm_scale_factor()
    im_poses[:, :3, 3] *= s_factor
    for img_pts3d in pts3d:
        img_pts3d *= s_factor
    if self.has_im_poses:
        for i in range(self.n_imgs):
            cam2world = im_poses[i]
            depth = geotrf(inv(cam2world), pts3d[i])[..., 2]
            self._set_depthmap(i, depth)
            self._set_pose(self.im_poses, i, cam2world)
            if im_focals[i] is not None:
                self._set_focal(i, im_focals[i])
    if self.verbose:
        print(' init loss =', float(self()))


def minimum_spanning_tree(imshapes, edges, pred_i, pred_j, conf_i, conf_j, im_conf, min_conf_thr, device, has_im_poses=True,
                          niter_PnP=10, verbose=True):
    n_imgs = len(imshapes)
    sparse_graph = -dict_to_sparse_graph(compute_edge_scores(map(i_j_ij, edges), conf_i, conf_j))
    msp = sp.csgraph.minimum_spanning_tree(sparse_graph).tocoo()
    pts3d = [None] * n_imgs
    todo = sorted(zip(-msp.data, msp.row, msp.col))
    im_poses = [None] * n_imgs
    im_focals = [None] * n_imgs

    score, i, j = todo.pop()       # strongest edge seeds the world frame (= camera i)
    if verbose:
        print(f' init edge ({i}*,{j}*) {score=}')
    i_j = edge_str(i, j)
    pts3d[i] = pred_i[i_j].clone()
    pts3d[j] = pred_j[i_j].clone()
    done = {i, j}
    if has_im_poses:
        im_poses[i] = torch.eye(4, device=device)
        im_focals[i] = estimate_focal(pred_i[i_j])

    msp_edges = [(i, j)]
    while todo:
        score, i, j = todo.pop()
        if im_focals[i] is None:
            im_focals[i] = estimate_focal(pred_i[i_j])      # (sic) the reference re-uses the previous i_j here
        if i in done:
            if verbose:
                print(f' init edge ({i},{j}*) {score=}')
            assert j not in done
            i_j = edge_str(i, j)
            s, R, T = rigid_points_registration(pred_i[i_j], pts3d[i], conf=conf_i[i_j])
            pts3d[j] = geotrf(sRT_to_4x4(s, R, T, device), pred_j[i_j])
            done.add(j)
            msp_edges.append((i, j))
            if has_im_poses and im_poses[i] is None:
                im_poses[i] = sRT_to_4x4(1, R, T, device)
        elif j in done:
            if verbose:
                print(f' init edge ({i}*,{j}) {score=}')
            assert i not in done
            i_j = edge_str(i, j)
            s, R, T = rigid_points_registration(pred_j[i_j], pts3d[j], conf=conf_j[i_j])
            pts3d[i] = geotrf(sRT_to_4x4(s, R, T, device), pred_i[i_j])
            done.add(i)
            msp_edges.append((i, j))
            if has_im_poses and im_poses[i] is None:
                im_poses[i] = sRT_to_4x4(1, R, T, device)
        else:
            todo.insert(0, (score, i, j))   # neither end is placed yet: retry later

    if has_im_poses:
        pair_scores = list(sparse_graph.values())
        edges_from_best_to_worse = np.array(list(sparse_graph.keys()))[np.argsort(pair_scores)]
        for i, j in edges_from_best_to_worse.tolist():
            if im_focals[i] is None:
                im_focals[i] = estimate_focal(pred_i[edge_str(i, j)])
        for i in range(n_imgs):
            if im_poses[i] is None:
                msk = im_conf[i] > min_conf_thr
                res = fast_pnp(pts3d[i], im_focals[i], msk=msk, device=device, niter_PnP=niter_PnP)
                if res:
                    im_focals[i], im_poses[i] = res
                    im_poses[i] = im_poses[i].to(device)
            if im_poses[i] is None:
                im_poses[i] = torch.eye(4, device=device)
        im_poses = torch.stack(im_poses)
    else:
        im_poses = im_focals = None
    return pts3d, msp_edges, im_focals, im_poses
