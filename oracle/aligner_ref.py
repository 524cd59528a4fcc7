"""ORACLE (test infrastructure only) -- CPU restatement of the reference's global aligner
hot loop: `PointCloudOptimizer.forward` + `global_alignment_loop`.

Follows (citations into /root/reference/dust3r/cloud_opt/):
  optimizer.py:188-201   forward: li + lj, 'l1' = un-squared Euclidean norm, weight = log(conf)
  optimizer.py:170-180,204-211  depth_to_pts3d / _fast_depthmap_to_pts3d
  optimizer.py:127-129,141-142  get_focals (exp(f/focal_break)), get_principal_points
  base_opt.py:143-155,178-195   get_adaptors, _get_poses (roma RigidUnitQuat, signed_expm1),
                                get_pw_norm_scale_factor, get_pw_scale, get_pw_poses
  base_opt.py:326-366    Adam(lr, betas=(0.9, 0.9)) + cosine/linear schedule, one step per iter
  commons.py:62-90       l1_dist / l2_dist / signed_expm1 / schedules
Quaternion -> rotation uses oracle/roma_ref.py (roma itself is absent: PARITY UNPINNED for that
one function; everything else is pinned by tests/test_oracle_pins.py against the unmodified
reference files and by tests/golden/aligner_*.pt generated from them).

The state is held as plain tensors with the reference's `state_dict(trainable=True)` key names
(pw_poses, pw_adaptors, im_poses, im_depthmaps, im_focals, im_pp).
"""
import math
import os
import sys

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
if os.path.dirname(_HERE) not in sys.path:
    sys.path.insert(0, os.path.dirname(_HERE))

from oracle.roma_ref import unitquat_to_rotmat  # noqa: E402


def signed_expm1(x):
    return torch.sign(x) * torch.expm1(torch.abs(x))


def cosine_schedule(t, lr_start, lr_end):
    return lr_end + (lr_start - lr_end) * (1 + np.cos(t * np.pi)) / 2


def linear_schedule(t, lr_start, lr_end):
    return lr_start + (lr_end - lr_start) * t


class AlignerRef:
    def __init__(self, dust3r_output, dist='l1', conf='log', base_scale=0.5, pw_break=20, focal_break=20,
                 dtype=torch.float32):
        v1, v2, p1, p2 = [dust3r_output[k] for k in ('view1', 'view2', 'pred1', 'pred2')]
        self.edges = [(int(i), int(j)) for i, j in zip(v1['idx'], v2['idx'])]
        self.n_imgs = max(max(e) for e in self.edges) + 1
        E = len(self.edges)
        pi, pj = p1['pts3d'].cpu(), p2['pts3d_in_other_view'].cpu()
        H, W = pi.shape[1:3]
        self.H, self.W, self.A = H, W, H * W
        self.dtype = dtype
        self.dist = dist
        trf = dict(log=torch.log, sqrt=torch.sqrt, m1=lambda x: x - 1, id=lambda x: x, none=lambda x: x)[conf]
        self.pred_i = pi.reshape(E, self.A, 3).to(dtype)
        self.pred_j = pj.reshape(E, self.A, 3).to(dtype)
        self.weight_i = trf(p1['conf'].cpu().float()).reshape(E, self.A).to(dtype)
        self.weight_j = trf(p2['conf'].cpu().float()).reshape(E, self.A).to(dtype)
        self.ei = torch.tensor([i for i, j in self.edges])
        self.ej = torch.tensor([j for i, j in self.edges])
        self.base_scale, self.pw_break, self.focal_break = base_scale, pw_break, focal_break
        self.norm_pw_scale = True
        vs, us = torch.meshgrid(torch.arange(H, dtype=dtype), torch.arange(W, dtype=dtype), indexing='ij')
        self.grid = torch.stack((us, vs), dim=-1).reshape(1, self.A, 2)      # xy_grid(W,H): [...,0]=col
        self.pp0 = torch.tensor([W / 2, H / 2], dtype=dtype)
        self.total_area_i = self.total_area_j = E * self.A
        self.params = None

    # ------------------------------------------------------------------ state
    def load_state(self, state, optimize_pp=False, allow_pw_adaptors=False):
        # optimizer.py:34: im_pp.requires_grad_(optimize_pp); base_opt.py:92: pw_adaptors.requires_grad_(allow_pw_adaptors)
        frozen = tuple(k for k, on in (('pw_adaptors', allow_pw_adaptors), ('im_pp', optimize_pp)) if not on)
        self.params = {k: state[k].detach().clone().to(self.dtype).requires_grad_(k not in frozen)
                       for k in ('pw_poses', 'pw_adaptors', 'im_poses', 'im_depthmaps', 'im_focals', 'im_pp')}
        return self

    def state(self):
        return {k: v.detach().clone() for k, v in self.params.items()}

    # ------------------------------------------------------------------ forward pieces
    def _poses(self, P):
        R = unitquat_to_rotmat(P[:, :4] / P[:, :4].norm(dim=-1, keepdim=True))
        return R, signed_expm1(P[:, 4:7])

    def pw_scale(self):
        s = self.params['pw_poses'][:, -1]
        sc = s.exp()
        if self.norm_pw_scale:
            sc = sc * (math.log(self.base_scale) - s.mean()).exp()
        return sc

    def im_poses(self):
        R, T = self._poses(self.params['im_poses'])
        M = torch.zeros((self.n_imgs, 4, 4), dtype=self.dtype)
        M[:, :3, :3], M[:, :3, 3], M[:, 3, 3] = R, T, 1
        return M

    def focals(self):
        return (self.params['im_focals'] / self.focal_break).exp()

    def pts3d(self):
        p = self.params
        Ri, Ti = self._poses(p['im_poses'])
        f = self.focals().unsqueeze(1)                                       # (n,1,1)
        pp = (self.pp0 + 10 * p['im_pp']).unsqueeze(1)                       # (n,1,2)
        depth = p['im_depthmaps'].exp().unsqueeze(-1)                        # (n,A,1)
        cam = torch.cat((depth * (self.grid - pp) / f, depth), dim=-1)       # (n,A,3)
        return cam @ Ri.transpose(1, 2) + Ti[:, None, :]

    def loss(self):
        p = self.params
        Re, Te = self._poses(p['pw_poses'])
        sc = self.pw_scale().view(-1, 1, 1)
        a = p['pw_adaptors']
        adapt = torch.cat((a[:, 0:1], a), dim=-1)
        if self.norm_pw_scale:
            adapt = adapt - adapt.mean(dim=1, keepdim=True)
        adapt = (adapt / self.pw_break).exp().unsqueeze(1)                   # (E,1,3)
        X = self.pts3d()
        sR, sT = sc * Re, (sc.view(-1, 1) * Te)[:, None, :]
        ai = (adapt * self.pred_i) @ sR.transpose(1, 2) + sT
        aj = (adapt * self.pred_j) @ sR.transpose(1, 2) + sT
        if self.dist == 'l1':
            di = (X[self.ei] - ai).norm(dim=-1)
            dj = (X[self.ej] - aj).norm(dim=-1)
        else:
            di = (X[self.ei] - ai).square().sum(dim=-1)
            dj = (X[self.ej] - aj).square().sum(dim=-1)
        return (di * self.weight_i).sum() / self.total_area_i + (dj * self.weight_j).sum() / self.total_area_j

    # ------------------------------------------------------------------ loop
    def run(self, niter=300, lr=0.01, schedule='cosine', lr_min=1e-6, callback=None, total=None, start=0):
        """`niter` Adam iterations. By default one whole schedule (the reference's global_alignment_loop, base_opt.py:326-366);
        `total` / `start` run iterations [start, start + niter) of a `total`-iteration schedule, keeping the Adam moments of the
        previous call when start > 0 (a long run cut into pieces gives the same trajectory as one call)."""
        total = niter if total is None else total
        params = [v for v in self.params.values() if v.requires_grad]
        if start == 0 or getattr(self, '_opt', None) is None:
            self._opt = torch.optim.Adam(params, lr=lr, betas=(0.9, 0.9))
        opt = self._opt
        losses = []
        for n in range(start, start + niter):
            t = n / total
            cur = cosine_schedule(t, lr, lr_min) if schedule == 'cosine' else linear_schedule(t, lr, lr_min)
            for grp in opt.param_groups:
                grp['lr'] = cur
            opt.zero_grad()
            loss = self.loss()
            loss.backward()
            opt.step()
            losses.append(float(loss))
            if callback is not None:
                callback(n, self)
        return losses

    def grads(self):
        """One forward/backward without a step: (loss, {name: grad}) -- used by the kernel tests."""
        for v in self.params.values():
            v.grad = None
        loss = self.loss()
        loss.backward()
        return float(loss), {k: v.grad.detach().clone() for k, v in self.params.items() if v.requires_grad}
