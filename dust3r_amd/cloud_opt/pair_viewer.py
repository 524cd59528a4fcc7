"""`PairViewer`: the no-optimisation "aligner" for exactly one symmetrised pair (BASELINE config 1; reference
`dust3r/cloud_opt/pair_viewer.py:18-127`). Same recipe -- Weiszfeld focal of each view's own pointmap, relative pose by PnP of a
view's pixels against its points seen from the other view, depth from the more confident direction -- evaluated by the GPU scene
bootstrap (csrc/bootstrap.hip through cloud_opt/bootstrap.py): two focal fits in one launch, both PnP problems batched, the two
depth maps in one launch. Solved lazily on first use, once the scene sits on its GPU (`global_aligner` moves it after building)."""
import numpy as np
import torch

from .base_opt import BasePCOptimizer
from .bootstrap import PairMaps


class PairViewer(BasePCOptimizer):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        assert self.is_symmetrized and self.n_edges == 2
        self.has_im_poses = True
        self._solution = None

    # ------------------------------------------------------------------ the solve
    @torch.no_grad()
    def _solved(self):
        if self._solution is not None:
            return self._solution
        maps = PairMaps(self)                                   # raises when the scene is not on a GPU
        dev = maps.dev
        e_of = {ij: e for e, ij in enumerate(self.edges)}
        mean_i, mean_j = maps.edge_conf_means()
        lead = [e_of[(i, 1 - i)] for i in (0, 1)]              # edge whose view 1 is image i
        confs = [float(mean_i[e] * mean_j[e]) for e in lead]
        focals = maps.weiszfeld_focals([(0, e) for e in lead])
        if self.verbose:
            for i in (0, 1):
                print(f'  - conf={confs[i]:.3} for edge {i}-{1 - i}')
        jobs = []
        for i in (0, 1):
            H, W = self.imshapes[i]
            e = e_of[(1 - i, i)]                                # image i's points in the frame of camera 1 - i: view 2 of that edge
            conf = self.im_conf[i].contiguous()
            jobs.append(dict(map=maps.map_addr(1, e), conf=conf.data_ptr(), G=np.eye(4)[:3], f=float(focals[i]), pp=(W / 2, H / 2),
                             thr=float(self.min_conf_thr), H=H, W=W, points=maps.preds[1][e][:H * W], confs=conf))
        rel = []
        for ok, w2c, _ in maps.solve_pnp(jobs, iterations=100):
            rel.append(np.linalg.inv(w2c) if ok else np.eye(4))   # camera i -> frame of camera 1 - i
        a = 0 if confs[0] > confs[1] else 1                       # the cloud lives in camera a's frame
        poses = [np.eye(4), np.eye(4)]
        poses[1 - a] = rel[1 - a]
        e = lead[a]
        rows = [None, None]
        rows[a] = np.array([0, 0, 1, 0.0])                        # z of pred_i[a_(1-a)]
        rows[1 - a] = np.linalg.inv(rel[1 - a])[2]                # z of pred_j[a_(1-a)] seen from camera 1 - a
        anchors = [None, None]
        anchors[a], anchors[1 - a] = (0, e), (1, e)
        depth = torch.empty((2, self.max_area), dtype=torch.float32, device=dev)
        maps.anchor_depth(anchors, rows, depth, take_log=False)
        self._solution = dict(
            im_poses=torch.tensor(np.stack(poses), dtype=torch.float32, device=dev),
            focals=torch.tensor(np.asarray(focals), dtype=torch.float32, device=dev),
            pp=torch.tensor([(w / 2, h / 2) for h, w in self.imshapes], dtype=torch.float32, device=dev),
            depth=[depth[i, :h * w].view(h, w) for i, (h, w) in enumerate(self.imshapes)])
        return self._solution

    def to(self, device, *a, **k):
        self._solution = None
        return super().to(device, *a, **k)

    # ------------------------------------------------------------------ the reference's getter surface
    def trainable_names(self):
        return []

    def _set_depthmap(self, idx, depth, force=False):
        if self.verbose:
            print('_set_depthmap is ignored in PairViewer')

    def get_depthmaps(self, raw=False):
        return list(self._solved()['depth'])

    def _set_focal(self, idx, focal, force=False):
        self._solved()['focals'][idx] = focal

    def get_focals(self):
        return self._solved()['focals']

    def get_known_focal_mask(self):
        return torch.tensor([True] * self.n_imgs)

    def get_principal_points(self):
        return self._solved()['pp']

    def get_intrinsics(self):
        s = self._solved()
        K = torch.zeros((self.n_imgs, 3, 3), device=s['focals'].device)
        K[:, 0, 0] = K[:, 1, 1] = s['focals']
        K[:, :2, 2] = s['pp']
        K[:, 2, 2] = 1
        return K

    def get_im_poses(self):
        return self._solved()['im_poses']

    def depth_to_pts3d(self):
        """World points of both views from (depth, intrinsics, pose): X = pose . (d (u - ppx) / f, d (v - ppy) / f, d)."""
        s = self._solved()
        out = []
        for i, (h, w) in enumerate(self.imshapes):
            d = s['depth'][i]
            v, u = torch.meshgrid(torch.arange(h, device=d.device, dtype=torch.float32), torch.arange(w, device=d.device, dtype=torch.float32),
                                  indexing='ij')
            f, (px, py) = s['focals'][i], s['pp'][i]
            cam = torch.stack((d * (u - px) / f, d * (v - py) / f, d), dim=-1)
            P = s['im_poses'][i]
            out.append(cam @ P[:3, :3].T + P[:3, 3])
        return out

    def get_pts3d(self, raw=False):
        return self.depth_to_pts3d()

    def compute_global_alignment(self, *a, **k):
        return float('nan')

    def forward(self):
        return float('nan')
