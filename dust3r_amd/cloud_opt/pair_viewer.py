"""`PairViewer` -- host-side mirror of the reference `dust3r/cloud_opt/pair_viewer.py:18-127`: the
no-optimisation "aligner" for exactly one symmetrised pair (BASELINE config 1). Focals by the
Weiszfeld estimator, relative pose by PnP-RANSAC (own solver instead of cv2), depth taken from the
more confident direction. Pure host code (numpy / torch CPU), as in the reference."""
import numpy as np
import torch
import torch.nn as nn

from ..post_process import estimate_focal_knowing_depth
from ..utils.geometry import depthmap_to_absolute_camera_coordinates, geotrf, inv
from .base_opt import BasePCOptimizer
from .commons import edge_str
from .pnp import solve_pnp_ransac


class PairViewer(BasePCOptimizer):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        assert self.is_symmetrized and self.n_edges == 2
        self.has_im_poses = True
        focals, pps, rel_poses, confs = [], [], [], []
        for i in range(self.n_imgs):
            conf = float(self.conf_i[edge_str(i, 1 - i)].mean() * self.conf_j[edge_str(i, 1 - i)].mean())
            if self.verbose:
                print(f'  - {conf=:.3} for edge {i}-{1 - i}')
            confs.append(conf)
            H, W = self.imshapes[i]
            pts3d = self.pred_i[edge_str(i, 1 - i)].cpu()
            pp = torch.tensor((W / 2, H / 2))
            focal = float(estimate_focal_knowing_depth(pts3d[None], pp, focal_mode='weiszfeld'))
            focals.append(focal)
            pps.append(pp)
            # pose of camera i in the frame of camera 1-i: PnP of image i's pixels against its points seen from 1-i
            pixels = np.mgrid[:W, :H].T.astype(np.float32)
            pts3d = self.pred_j[edge_str(1 - i, i)].cpu().numpy()
            assert pts3d.shape[:2] == (H, W)
            msk = self.get_masks()[i].cpu().numpy()
            K = np.float32([(focal, 0, pp[0]), (0, focal, pp[1]), (0, 0, 1)])
            pose = np.eye(4)
            try:
                ok, R, T, _ = solve_pnp_ransac(pts3d[msk], pixels[msk], K, iterations=100, reproj_err=5)
                if ok:
                    pose = inv(np.r_[np.c_[R, T], [(0, 0, 0, 1)]])
            except Exception:
                pose = np.eye(4)
            rel_poses.append(torch.from_numpy(pose.astype(np.float32)))
        if confs[0] > confs[1]:   # cloud expressed in camera 0
            im_poses = [torch.eye(4), rel_poses[1]]
            depth = [self.pred_i['0_1'][..., 2].cpu(), geotrf(inv(rel_poses[1]), self.pred_j['0_1'].cpu())[..., 2]]
        else:                     # cloud expressed in camera 1
            im_poses = [rel_poses[0], torch.eye(4)]
            depth = [geotrf(inv(rel_poses[0]), self.pred_j['1_0'].cpu())[..., 2], self.pred_i['1_0'][..., 2].cpu()]
        self.im_poses = nn.Parameter(torch.stack(im_poses, dim=0), requires_grad=False)
        self.focals = nn.Parameter(torch.tensor(focals), requires_grad=False)
        self.pp = nn.Parameter(torch.stack(pps, dim=0), requires_grad=False)
        self.depth = nn.ParameterList([nn.Parameter(d, requires_grad=False) for d in depth])
        for p in self.parameters():
            p.requires_grad = False

    def trainable_names(self):
        return []

    def _set_depthmap(self, idx, depth, force=False):
        if self.verbose:
            print('_set_depthmap is ignored in PairViewer')

    def get_depthmaps(self, raw=False):
        return [d.to(self.device) for d in self.depth]

    def _set_focal(self, idx, focal, force=False):
        self.focals[idx] = focal

    def get_focals(self):
        return self.focals

    def get_known_focal_mask(self):
        return torch.tensor([True] * len(self.focals))

    def get_principal_points(self):
        return self.pp

    def get_intrinsics(self):
        focals, pps = self.get_focals(), self.get_principal_points()
        K = torch.zeros((len(focals), 3, 3), device=self.device)
        for i in range(len(focals)):
            K[i, 0, 0] = K[i, 1, 1] = focals[i]
            K[i, :2, 2] = pps[i]
            K[i, 2, 2] = 1
        return K

    def get_im_poses(self):
        return self.im_poses

    def depth_to_pts3d(self):
        out = []
        for d, K, pose in zip(self.depth, self.get_intrinsics(), self.get_im_poses()):
            pts, _ = depthmap_to_absolute_camera_coordinates(d.cpu().numpy(), K.cpu().numpy(), pose.cpu().numpy())
            out.append(torch.from_numpy(pts).to(device=self.device))
        return out

    def get_pts3d(self, raw=False):
        return self.depth_to_pts3d()

    def compute_global_alignment(self, *a, **k):
        return float('nan')

    def forward(self):
        return float('nan')
