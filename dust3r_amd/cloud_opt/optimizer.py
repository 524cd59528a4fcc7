"""`PointCloudOptimizer` -- host-side mirror of the reference `dust3r/cloud_opt/optimizer.py:16-237`.

Holds the reference's parameter tensors under the reference's names and parameterisation
(`im_depthmaps` log-depth (n, max_area), `im_poses` (n, 7) quat XYZW + signed-log translation,
`im_focals` (n, 1) = focal_break * log f, `im_pp` (n, 2), `pw_poses` (E, 8)) and exposes the same
getters / presets. `forward()` (the loss) and the optimisation loop are evaluated by the fused HIP
aligner, which reads and updates these tensors in place.
"""
import ctypes as C

import numpy as np
import torch
import torch.nn as nn

from .. import _lib
from .._lib import check, current_stream, lib, ptr
from ..utils.device import to_cpu, to_numpy
from ..utils.geometry import geotrf, xy_grid
from .base_opt import BasePCOptimizer, _ravel_hw


class PointCloudOptimizer(BasePCOptimizer):
    def __init__(self, *args, optimize_pp=False, focal_break=20, **kwargs):
        super().__init__(*args, **kwargs)
        self.has_im_poses = True
        self.focal_break = focal_break
        n = self.n_imgs
        # same initial distributions as optimizer.py:29-34
        areas = torch.tensor([H * W for H, W in self.imshapes])
        depth0 = torch.randn((n, self.max_area)).div_(10).sub_(3)                            # one draw for all images, scaled in place (79 MB at 100 views: no second and third copy to page in)
        if int(areas.min()) < self.max_area:
            depth0.mul_(torch.arange(self.max_area)[None, :] < areas[:, None])                  # zero in the padding
        self.im_depthmaps = nn.Parameter(depth0)
        self.im_poses = nn.Parameter(torch.stack([self.rand_pose(self.POSE_DIM) for _ in range(n)]).float())
        self.im_focals = nn.Parameter(torch.tensor([[self.focal_break * np.log(max(H, W))] for H, W in self.imshapes], dtype=torch.float32))
        self.im_pp = nn.Parameter(torch.zeros((n, 2)), requires_grad=bool(optimize_pp))       # optimizer.py:34: im_pp.requires_grad_(optimize_pp)
        self.imshape = self.imshapes[0]
        self.register_buffer('_pp', torch.tensor([(w / 2, h / 2) for h, w in self.imshapes], dtype=torch.float32))
        self._grid_cache = None        # (n, max_area, 2) pixel grid of depth_to_pts3d, built on first use on the scene's device
        self.register_buffer('_ei', torch.tensor([i for i, j in self.edges]))
        self.register_buffer('_ej', torch.tensor([j for i, j in self.edges]))
        im_areas = [h * w for h, w in self.imshapes]
        self.total_area_i = sum(im_areas[i] for i, j in self.edges)
        self.total_area_j = sum(im_areas[j] for i, j in self.edges)

    def trainable_names(self):
        return [k for k in ('pw_poses', 'pw_adaptors', 'im_depthmaps', 'im_poses', 'im_focals', 'im_pp') if getattr(self, k).requires_grad]

    # ------------------------------------------------------------------ presets (optimizer.py:63-125)
    def _check_all_imgs_are_selected(self, msk):
        assert np.all(self._get_msk_indices(msk) == np.arange(self.n_imgs)), 'incomplete mask!'

    def _get_msk_indices(self, msk):
        if msk is None:
            return range(self.n_imgs)
        if isinstance(msk, int):
            return [msk]
        if isinstance(msk, (tuple, list)):
            return self._get_msk_indices(np.array(msk))
        if msk.dtype in (bool, torch.bool, np.bool_):
            assert len(msk) == self.n_imgs
            return np.where(msk)[0]
        if np.issubdtype(msk.dtype, np.integer):
            return msk
        raise ValueError(f'bad {msk=}')

    def preset_pose(self, known_poses, pose_msk=None):
        self._check_all_imgs_are_selected(pose_msk)
        if isinstance(known_poses, torch.Tensor) and known_poses.ndim == 2:
            known_poses = [known_poses]
        for idx, pose in zip(self._get_msk_indices(pose_msk), known_poses):
            if self.verbose:
                print(f' (setting pose #{idx} = {pose[:3, 3]})')
            assert self.im_poses.requires_grad, 'it must be True at this point, otherwise no modification occurs'
            self._set_pose(self.im_poses, idx, torch.as_tensor(pose))
        self.im_poses.requires_grad_(False)
        self.norm_pw_scale = False
        self._destroy_engine()

    def preset_focal(self, known_focals, msk=None):
        self._check_all_imgs_are_selected(msk)
        for idx, focal in zip(self._get_msk_indices(msk), known_focals):
            if self.verbose:
                print(f' (setting focal #{idx} = {focal})')
            assert self.im_focals.requires_grad
            self._set_focal(idx, focal)
        self.im_focals.requires_grad_(False)
        self._destroy_engine()

    def preset_principal_point(self, known_pp, msk=None):
        self._check_all_imgs_are_selected(msk)
        for idx, pp in zip(self._get_msk_indices(msk), known_pp):
            if self.verbose:
                print(f' (setting principal point #{idx} = {pp})')
            self._set_principal_point(idx, pp, force=True)
        self.im_pp.requires_grad_(False)
        self._destroy_engine()

    def _set_focal(self, idx, focal, force=False):
        if self.im_focals.requires_grad or force:
            with torch.no_grad():
                self.im_focals.data[idx] = float(self.focal_break * np.log(float(focal)))
        return self.im_focals[idx]

    def _set_principal_point(self, idx, pp, force=False):
        H, W = self.imshapes[idx]
        if self.im_pp.requires_grad or force:
            with torch.no_grad():
                self.im_pp.data[idx] = torch.as_tensor((np.asarray(to_numpy(pp), np.float32) - (W / 2, H / 2)) / 10, dtype=torch.float32)
        return self.im_pp[idx]

    def _set_depthmap(self, idx, depth, force=False):
        depth = _ravel_hw(depth, self.max_area)
        if self.im_depthmaps.requires_grad or force:
            with torch.no_grad():
                self.im_depthmaps.data[idx] = depth.log().nan_to_num(neginf=0).to(self.im_depthmaps.device)
        return self.im_depthmaps[idx]

    # ------------------------------------------------------------------ getters (optimizer.py:127-186)
    def get_focals(self):
        return (self.im_focals / self.focal_break).exp()

    def get_known_focal_mask(self):
        return torch.tensor([not self.im_focals.requires_grad] * self.n_imgs)

    def get_principal_points(self):
        return self._pp + 10 * self.im_pp

    def get_intrinsics(self):
        K = torch.zeros((self.n_imgs, 3, 3), device=self.device)
        focals = self.get_focals().flatten()
        K[:, 0, 0] = K[:, 1, 1] = focals
        K[:, :2, 2] = self.get_principal_points()
        K[:, 2, 2] = 1
        return K

    def get_im_poses(self):
        return self._get_poses(self.im_poses)

    def get_depthmaps(self, raw=False):
        res = self.im_depthmaps.exp()
        if not raw:
            res = [dm[:h * w].view(h, w) for dm, (h, w) in zip(res, self.imshapes)]
        return res

    @property
    def _grid(self):
        g = self._grid_cache
        if g is None or g.device != self.device:
            dev = self.device
            per_shape = {hw: _ravel_hw(xy_grid(hw[1], hw[0], device=dev).float(), self.max_area) for hw in set(self.imshapes)}
            g = self._grid_cache = torch.stack([per_shape[hw] for hw in self.imshapes])
        return g

    def depth_to_pts3d(self):
        focals = self.get_focals().unsqueeze(1)                 # (n,1,1)
        pp = self.get_principal_points().unsqueeze(1)           # (n,1,2)
        depth = self.get_depthmaps(raw=True).unsqueeze(-1)      # (n,A,1)
        rel = torch.cat((depth * (self._grid - pp) / focals, depth), dim=-1)
        return geotrf(self.get_im_poses(), rel)

    # ------------------------------------------------------------------ engine binding
    def _ensure_engine(self):
        _lib.require_device()
        if self.device.type != 'cuda':
            raise _lib.D3RError('the aligner is not on a GPU: call .to("cuda") (dust3r_amd has no CPU execution path)')
        sig = (self.norm_pw_scale, self.im_poses.requires_grad, self.im_focals.requires_grad, self.im_pp.requires_grad, self.pw_adaptors.requires_grad, self.dist_name,
               tuple(getattr(self, k).data_ptr() for k in ('pw_poses', 'im_depthmaps', 'im_poses', 'im_focals', 'im_pp', 'pw_adaptors')))
        if self._engine is not None and sig == self._engine_sig:
            return self._engine
        self._destroy_engine()
        for k in ('_stacked_pred_i', '_stacked_pred_j', '_weight_i', '_weight_j', 'pw_poses', 'pw_adaptors', 'im_poses',
                  'im_depthmaps', 'im_focals', 'im_pp'):
            t = getattr(self, k)
            assert t.is_cuda and t.is_contiguous() and t.dtype == torch.float32, f'{k} must be a contiguous fp32 CUDA tensor'
        n, E = self.n_imgs, self.n_edges
        arr = lambda v: (C.c_int * len(v))(*v)  # noqa: E731
        ei, ej = arr([i for i, j in self.edges]), arr([j for i, j in self.edges])
        hh, ww = arr([h for h, w in self.imshapes]), arr([w for h, w in self.imshapes])
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            check(lib.d3r_aligner_create(C.byref(h), n, E, ei, ej, hh, ww, self.max_area, ptr(self._stacked_pred_i),
                                         ptr(self._stacked_pred_j), ptr(self._weight_i), ptr(self._weight_j), ptr(self.pw_poses.data),
                                         ptr(self.pw_adaptors.data), ptr(self.im_poses.data), ptr(self.im_depthmaps.data),
                                         ptr(self.im_focals.data), ptr(self.im_pp.data), float(self.base_scale), float(self.pw_break),
                                         float(self.focal_break), int(self.dist_name == 'l2'), int(self.norm_pw_scale),
                                         int(self.im_poses.requires_grad), int(self.im_focals.requires_grad), 1024, current_stream()), 'aligner_create')
            check(lib.d3r_aligner_set_option(h, 3, int(self.im_pp.requires_grad)), 'set_option(optimize_pp)')
            check(lib.d3r_aligner_set_option(h, 4, int(self.pw_adaptors.requires_grad)), 'set_option(allow_pw_adaptors)')
        self._engine, self._engine_sig = h, sig
        return h

    @torch.no_grad()
    def forward(self):
        """The alignment loss (optimizer.py:188-201), evaluated by the engine (no parameter update)."""
        eng = self._ensure_engine()
        loss = torch.empty(1, dtype=torch.float32, device=self.device)
        check(lib.d3r_aligner_loss_grad(eng, ptr(loss), None, None, None, None, None, None, current_stream()), 'aligner_loss')
        return loss[0]

    @torch.no_grad()
    def loss_and_grads(self):
        """(loss, {name: grad}) of one forward/backward without a step -- the engine's analytic gradients."""
        eng = self._ensure_engine()
        loss = torch.empty(1, dtype=torch.float32, device=self.device)
        g = {k: torch.zeros_like(getattr(self, k).data) for k in ('pw_poses', 'im_poses', 'im_depthmaps', 'im_focals', 'im_pp', 'pw_adaptors')}
        check(lib.d3r_aligner_loss_grad(eng, ptr(loss), ptr(g['pw_poses']), ptr(g['im_poses']), ptr(g['im_depthmaps']),
                                        ptr(g['im_focals']), ptr(g['im_pp']), ptr(g['pw_adaptors']), current_stream()), 'aligner_loss_grad')
        return loss[0], g

    def set_reduction(self, use_dpp=True):
        check(lib.d3r_aligner_set_option(self._ensure_engine(), 1, int(use_dpp)), 'set_option')
