"""Global alignment -- host-side mirror of the reference `dust3r/cloud_opt/base_opt.py`
(`BasePCOptimizer`, `global_alignment_loop`).

Same constructor keywords, attributes (`edges, imshapes, imsizes, im_conf, pred_i, pred_j, conf_i,
conf_j, pw_poses, pw_adaptors, min_conf_thr, conf_trf, is_symmetrized, n_imgs, n_edges,
str_edges`), getters and `compute_global_alignment(init, niter_PnP, lr, niter, schedule, lr_min)`.
What differs is WHERE the optimisation runs: the reference builds an autograd graph of ~25 kernels
per iteration and steps torch.optim.Adam (base_opt.py:326-366); here `global_alignment_loop` hands
the parameter tensors to the fused HIP aligner (csrc/aligner.hip, C ABI `d3r_aligner_*`), which
updates them in place. There is no CPU execution path for the loop.
"""
import ctypes as C

import numpy as np
import torch
import torch.nn as nn
import tqdm

from .. import _lib
from .._lib import check, current_stream, lib, ptr
from ..utils.geometry import inv
from ..utils.rigid import quat_translation_to_homogeneous, rotmat_to_unitquat
from . import init_im_poses as init_fun
from .commons import (cosine_schedule, edge_str, get_conf_trf, get_imshapes, linear_schedule, signed_expm1,
                      signed_log1p)


def _ravel_hw(tensor, fill=0):
    """(H, W, ...) -> (H*W, ...) zero padded to `fill` rows (reference optimizer.py:231-237)."""
    tensor = tensor.reshape((tensor.shape[0] * tensor.shape[1],) + tuple(tensor.shape[2:]))
    if len(tensor) < fill:
        tensor = torch.cat((tensor, tensor.new_zeros((fill - len(tensor),) + tuple(tensor.shape[1:]))))
    return tensor


class _EdgeView:
    """dict-like access `view['i_j'] -> (H, W, ...)` into a stacked (E, max_area, ...) tensor."""

    def __init__(self, owner, attr, side):
        self._o, self._attr, self._side = owner, attr, side

    def _index(self, key):
        return self._o._edge_index[key]

    def __getitem__(self, key):
        e = self._index(key)
        i, j = self._o.edges[e]
        h, w = self._o.imshapes[i if self._side == 0 else j]
        t = getattr(self._o, self._attr)[e]
        return t[:h * w].view((h, w) + tuple(t.shape[1:]))

    def __contains__(self, key):
        return key in self._o._edge_index

    def keys(self):
        return self._o._edge_index.keys()

    def items(self):
        return [(k, self[k]) for k in self.keys()]

    def values(self):
        return [self[k] for k in self.keys()]

    def __len__(self):
        return len(self._o._edge_index)


class BasePCOptimizer(nn.Module):
    """Optimize a global scene given pairwise observations. Nodes: images; edges: (pred1, pred2)."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        self._init_from_views(*args, **kwargs)

    def _init_from_views(self, view1, view2, pred1, pred2, dist='l1', conf='log', min_conf_thr=3, base_scale=0.5,
                         allow_pw_adaptors=False, pw_break=20, rand_pose=torch.randn, iterationsCount=None, verbose=True):
        if dist not in ('l1', 'l2'):
            raise KeyError(dist)
        if not isinstance(view1['idx'], list):
            view1['idx'] = view1['idx'].tolist()
        if not isinstance(view2['idx'], list):
            view2['idx'] = view2['idx'].tolist()
        self.edges = [(int(i), int(j)) for i, j in zip(view1['idx'], view2['idx'])]
        self.is_symmetrized = set(self.edges) == {(j, i) for i, j in self.edges}
        self.dist_name = dist
        self.verbose = verbose
        self.n_imgs = self._check_edges()
        self._edge_index = {edge_str(i, j): e for e, (i, j) in enumerate(self.edges)}

        pred1_pts, pred2_pts = pred1['pts3d'], pred2['pts3d_in_other_view']
        pred1_conf, pred2_conf = pred1['conf'], pred2['conf']
        self.imshapes = get_imshapes(self.edges, pred1_pts, pred2_pts)
        im_areas = [h * w for h, w in self.imshapes]
        self.max_area = max(im_areas)
        assert all(a % 4 == 0 for a in im_areas), 'image areas must be multiples of 4'

        def stack(seq):
            if isinstance(seq, torch.Tensor) and seq.shape[1] * seq.shape[2] == self.max_area:
                return seq.detach().float().reshape((seq.shape[0], self.max_area) + tuple(seq.shape[3:])).contiguous()
            return torch.stack([_ravel_hw(torch.as_tensor(p).detach().float(), self.max_area) for p in seq]).contiguous()

        self.register_buffer('_stacked_pred_i', stack(pred1_pts))
        self.register_buffer('_stacked_pred_j', stack(pred2_pts))
        self.register_buffer('_conf_i', stack(pred1_conf))
        self.register_buffer('_conf_j', stack(pred2_conf))
        self.pred_i, self.pred_j = _EdgeView(self, '_stacked_pred_i', 0), _EdgeView(self, '_stacked_pred_j', 1)
        self.conf_i, self.conf_j = _EdgeView(self, '_conf_i', 0), _EdgeView(self, '_conf_j', 1)

        self.min_conf_thr = min_conf_thr
        self.conf_mode = conf
        self.conf_trf = get_conf_trf(conf)
        self.im_conf = self._compute_img_conf()

        # pre-computed pixel weights (zero in the padding, like ParameterStack(fill=max_area))
        def weights(c, side):
            w = self.conf_trf(c.clamp_min(1e-30)) if conf == 'log' else self.conf_trf(c)
            for e, (i, j) in enumerate(self.edges):
                a = im_areas[i if side == 0 else j]
                if a < self.max_area:
                    w[e, a:] = 0
            return w.contiguous()
        self.register_buffer('_weight_i', weights(self._conf_i.clone(), 0))
        self.register_buffer('_weight_j', weights(self._conf_j.clone(), 1))

        self.base_scale = base_scale
        self.norm_pw_scale = True
        self.pw_break = pw_break
        self.POSE_DIM = 7
        self.pw_poses = nn.Parameter(rand_pose((self.n_edges, 1 + self.POSE_DIM)).float())
        self.pw_adaptors = nn.Parameter(torch.zeros((self.n_edges, 2)), requires_grad=bool(allow_pw_adaptors))     # base_opt.py:92
        self.has_im_poses = False
        self.rand_pose = rand_pose

        self.imgs = None
        if 'img' in view1 and 'img' in view2:
            from ..utils.image import rgb
            imgs = [torch.zeros((3,) + hw) for hw in self.imshapes]
            for v in range(len(self.edges)):
                imgs[view1['idx'][v]] = view1['img'][v]
                imgs[view2['idx'][v]] = view2['img'][v]
            self.imgs = rgb(imgs)
        self._engine = None
        self._engine_sig = None

    # ------------------------------------------------------------------ bookkeeping
    @property
    def n_edges(self):
        return len(self.edges)

    @property
    def str_edges(self):
        return [edge_str(i, j) for i, j in self.edges]

    @property
    def imsizes(self):
        return [(w, h) for h, w in self.imshapes]

    @property
    def device(self):
        return self.pw_poses.device

    def _check_edges(self):
        indices = sorted({i for edge in self.edges for i in edge})
        assert indices == list(range(len(indices))), 'bad pair indices: missing values '
        return len(indices)

    @torch.no_grad()
    def _compute_img_conf(self):
        """Per image, the pixel-wise maximum of the confidences of every edge side that shows it (base_opt.py:116-123 of the reference):
        two scatter-max passes over the stacked (E, max_area) confidences instead of 2 E small launches."""
        dev = self._conf_i.device
        acc = torch.zeros((self.n_imgs, self.max_area), dtype=torch.float32, device=dev)
        ei = torch.tensor([i for i, j in self.edges], device=dev)
        ej = torch.tensor([j for i, j in self.edges], device=dev)
        acc.index_reduce_(0, ei, self._conf_i, 'amax', include_self=True)
        acc.index_reduce_(0, ej, self._conf_j, 'amax', include_self=True)
        return [acc[i, :h * w].view(h, w) for i, (h, w) in enumerate(self.imshapes)]

    _TRAINABLE_KEYS = ('pw_poses', 'pw_adaptors', 'im_depthmaps', 'im_poses', 'im_focals', 'im_pp')

    def state_dict(self, trainable=True):
        if trainable:
            out = {k: getattr(self, k).detach().clone() for k in self._TRAINABLE_KEYS if hasattr(self, k)}
            out.update({f'im_conf.{i}': c.clone() for i, c in enumerate(self.im_conf)})
            return out
        return {k: getattr(self, k) for k in ('_stacked_pred_i', '_stacked_pred_j', '_weight_i', '_weight_j')}

    @torch.no_grad()
    def load_state_dict(self, data, **kw):
        for k, v in data.items():
            if k in self._TRAINABLE_KEYS and hasattr(self, k):
                getattr(self, k).data.copy_(torch.as_tensor(v).to(getattr(self, k).device).reshape(getattr(self, k).shape))
            elif k.startswith('im_conf.'):
                self.im_conf[int(k.split('.')[1])].copy_(torch.as_tensor(v))
        return self

    def to(self, device, *a, **k):
        self._destroy_engine()
        super().to(device, *a, **k)
        self.im_conf = [c.to(device) for c in self.im_conf]
        return self

    # ------------------------------------------------------------------ parameter access (cam-to-world)
    def get_adaptors(self):
        adapt = self.pw_adaptors
        adapt = torch.cat((adapt[:, 0:1], adapt), dim=-1)
        if self.norm_pw_scale:
            adapt = adapt - adapt.mean(dim=1, keepdim=True)
        return (adapt / self.pw_break).exp()

    def _get_poses(self, poses):
        return quat_translation_to_homogeneous(poses[:, :4], signed_expm1(poses[:, 4:7]))

    def _set_pose(self, poses, idx, R, T=None, scale=None, force=False):
        pose = poses[idx]
        if not (poses.requires_grad or force):
            return pose
        if R is not None and tuple(R.shape) == (4, 4):
            assert T is None
            T, R = R[:3, 3], R[:3, :3]
        with torch.no_grad():
            if R is not None:
                poses.data[idx, 0:4] = rotmat_to_unitquat(torch.as_tensor(R).float()).to(poses.device)
            if T is not None:
                poses.data[idx, 4:7] = signed_log1p(torch.as_tensor(T).float().to(poses.device) / (scale or 1))
            if scale is not None:
                assert poses.shape[-1] in (8, 13)
                poses.data[idx, -1] = float(np.log(float(scale)))
        return pose

    def get_pw_norm_scale_factor(self):
        if self.norm_pw_scale:
            return (np.log(self.base_scale) - self.pw_poses[:, -1].mean()).exp()
        return 1

    def get_pw_scale(self):
        return self.pw_poses[:, -1].exp() * self.get_pw_norm_scale_factor()

    def get_pw_poses(self):
        RT = self._get_poses(self.pw_poses)
        scaled = RT.clone()
        scaled[:, :3] *= self.get_pw_scale().view(-1, 1, 1)
        return scaled

    def get_masks(self):
        return [(conf > self.min_conf_thr) for conf in self.im_conf]

    def get_conf(self, mode=None):
        trf = self.conf_trf if mode is None else get_conf_trf(mode)
        return [trf(c) for c in self.im_conf]

    def depth_to_pts3d(self):
        raise NotImplementedError()

    def get_pts3d(self, raw=False):
        res = self.depth_to_pts3d()
        if not raw:
            res = [dm[:h * w].view(h, w, 3) for dm, (h, w) in zip(res, self.imshapes)]
        return res

    def get_focals(self):
        raise NotImplementedError()

    def get_im_poses(self):
        raise NotImplementedError()

    def get_depthmaps(self, raw=False):
        raise NotImplementedError()

    def get_intrinsics(self):
        raise NotImplementedError()

    @torch.no_grad()
    def clean_pointcloud(self, **kw):
        if self.device.type != 'cuda':
            raise _lib.D3RError('clean_pointcloud runs on the GPU (dust3r_amd has no CPU execution path)')
        cams = inv(self.get_im_poses())
        new_confs = clean_pointcloud_hip(self.im_conf, self.get_intrinsics(), cams, self.get_depthmaps(), self.get_pts3d(),
                                         tol=kw.get('tol', 0.001), bad_conf=kw.get('bad_conf', 0))
        for i, c in enumerate(new_confs):
            self.im_conf[i][:] = c
        return self

    def mask_sky(self):
        raise NotImplementedError('sky segmentation (cv2) belongs to the visualisation layer, outside this engine')

    def show(self, *a, **k):
        raise NotImplementedError('trimesh visualisation is outside this engine; export get_pts3d()/get_im_poses() instead')

    # ------------------------------------------------------------------ engine
    def _engine_signature(self):
        return None

    def _destroy_engine(self):
        if getattr(self, '_engine', None) is not None:
            lib.d3r_aligner_destroy(self._engine)
            self._engine = None

    def __del__(self):
        try:
            self._destroy_engine()
        except Exception:
            pass

    def forward(self, ret_details=False):
        raise NotImplementedError()

    def compute_global_alignment(self, init=None, niter_PnP=10, group=None, **kw):
        """`group` (new; the reference's loop is single-device): a torch.distributed process group (or True for the default one) whose ranks each hold this
        scene -- the return value of `inference_sharded` -- and then share the loop: every rank owns a contiguous range of images (global_alignment_loop_sharded).
        The initialisation runs replicated (it is deterministic); poses, focals and the final loss are the same on every rank, bit for bit what one GPU computes."""
        if init is None:
            pass
        elif init in ('msp', 'mst'):
            init_fun.init_minimum_spanning_tree(self, niter_PnP=niter_PnP)
        elif init == 'known_poses':
            init_fun.init_from_known_poses(self, min_conf_thr=self.min_conf_thr, niter_PnP=niter_PnP)
        else:
            raise ValueError(f'bad value for {init=}')
        if group is not None and group is not False:
            return global_alignment_loop_sharded(self, group=None if group is True else group, **kw)
        return global_alignment_loop(self, **kw)


def global_alignment_loop(net, lr=0.01, niter=300, schedule='cosine', lr_min=1e-6):
    """Mirror of base_opt.py:326-349: Adam(lr, betas=(0.9, 0.9)), lr scheduled per iteration; returns the
    loss of the last iteration. The iterations run inside the fused HIP aligner."""
    if schedule not in ('cosine', 'linear'):
        raise ValueError(f'bad lr {schedule=}')
    if niter <= 0:
        return float('inf')
    verbose = net.verbose
    if verbose:
        print('Global alignement - optimizing for:')
        print(net.trainable_names())
    eng = net._ensure_engine()
    check(lib.d3r_aligner_set_option(eng, 2, 0), 'reset adam')          # a fresh optimiser per call, as in the reference
    sched = 0 if schedule == 'cosine' else 1
    chunk = min(niter, 50 if verbose else 1024)
    losses = torch.empty(chunk, dtype=torch.float32, device=net.device)
    loss = float('inf')
    bar = tqdm.tqdm(total=niter) if verbose else None
    done = 0
    while done < niter:
        k = min(chunk, niter - done)
        check(lib.d3r_aligner_run(eng, k, done, niter, float(lr), float(lr_min), sched, ptr(losses), current_stream()), 'aligner_run')
        done += k
        if verbose or done >= niter:
            loss = float(losses[k - 1])                                    # the only host synchronisation
        if bar is not None:
            t = (done - 1) / niter
            cur = cosine_schedule(t, lr, lr_min) if schedule == 'cosine' else linear_schedule(t, lr, lr_min)
            bar.set_postfix_str(f'lr={cur:g} loss={loss:g}')
            bar.update(k)
    if bar is not None:
        bar.close()
    return loss


def image_ranges(edges, imshapes, world):
    """Contiguous image ranges [(first, count)] * world with balanced work: the main pass of an image costs its area times (the edge sides projected onto it + 1.5
    for its own depth map and Adam state: 32 B per edge-side pixel against 24 + 24 B per pixel). Ranks beyond the number of images get empty ranges."""
    n = len(imshapes)
    sides = [0] * n
    for i, j in edges:
        sides[i] += 1
        sides[j] += 1
    cost = [(sides[k] + 1.5) * imshapes[k][0] * imshapes[k][1] for k in range(n)]
    cum = [0.0]
    for c in cost:
        cum.append(cum[-1] + c)
    cuts = [0]
    for r in range(1, world):
        target = cum[-1] * r / world
        k = cuts[-1]
        while k < n and cum[k + 1] <= target:
            k += 1
        if k < n and target - cum[k] > cum[k + 1] - target:       # the nearer of the two neighbouring cuts
            k += 1
        cuts.append(max(k, cuts[-1]))
    cuts.append(n)
    return [(cuts[r], cuts[r + 1] - cuts[r]) for r in range(world)]


@torch.no_grad()
def global_alignment_loop_sharded(net, group=None, lr=0.01, niter=300, schedule='cosine', lr_min=1e-6):
    """global_alignment_loop over the ranks of `group` (one process per GPU; every rank holds the whole scene). Rank r runs the main pass of ITS images
    (include/dust3r_hip.h, d3r_aligner_set_image_range / step_begin / step_end); the reduced fp64 sums ((2 E + n) x 16 doubles: 160 KB at 100 views / 600 edges) are
    all-reduced once per iteration, and the pose / focal step runs replicated. Every partial record belongs to one image, i.e. to one rank: the other ranks add exact
    zeros, so losses and parameters are bit-identical to the single-GPU loop for any number of ranks. Start: all six parameter tensors are broadcast from rank 0 (a
    random `init=None` start differs between processes, for frozen tensors too); end: every rank receives the other ranks' rows of im_depthmaps."""
    import torch.distributed as dist
    if schedule not in ('cosine', 'linear'):
        raise ValueError(f'bad lr {schedule=}')
    if niter <= 0:
        return float('inf')
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    # ALL six parameter tensors, frozen ones included: a frozen tensor without a preset still holds its per-process random start (init=None), and the replicated
    # pose / focal step would then diverge silently between ranks (a few KB apart from the depth maps)
    for name in net._TRAINABLE_KEYS:
        dist.broadcast(getattr(net, name).data, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    eng = net._ensure_engine()
    ranges = image_ranges(net.edges, net.imshapes, world)
    first, count = ranges[rank]
    check(lib.d3r_aligner_set_option(eng, 2, 0), 'reset adam')
    check(lib.d3r_aligner_set_image_range(eng, first, count), 'set_image_range')
    red_ptr, red_n = C.c_void_p(), C.c_longlong()
    check(lib.d3r_aligner_reduced_sums(eng, C.byref(red_ptr), C.byref(red_n)), 'reduced_sums')
    # the engine's reduction buffer seen as a tensor (no copy): the collective works on it in place
    red = _device_view(red_ptr.value, int(red_n.value), torch.float64, net.device)
    sched = 0 if schedule == 'cosine' else 1
    cap = int(getattr(net, '_engine_max_iters', 1024))
    losses = torch.empty(min(niter, cap), dtype=torch.float32, device=net.device)
    done, loss = 0, float('inf')
    try:
        while done < niter:
            k_run = min(cap, niter - done)
            for k in range(k_run):
                check(lib.d3r_aligner_step_begin(eng, k, done, niter, float(lr), float(lr_min), sched, current_stream()), 'aligner_step_begin')
                dist.all_reduce(red, op=dist.ReduceOp.SUM, group=group)
                check(lib.d3r_aligner_step_end(eng, k, done, niter, float(lr), float(lr_min), sched, current_stream()), 'aligner_step_end')
            check(lib.d3r_aligner_read_losses(eng, k_run, ptr(losses), current_stream()), 'aligner_read_losses')
            done += k_run
            loss = float(losses[k_run - 1])
    finally:
        # never raise from here: an error of the loop above must reach the caller as itself (the other ranks see it as a collective timeout)
        rc = lib.d3r_aligner_set_image_range(eng, 0, net.n_imgs)
        if rc != 0:
            import logging
            logging.getLogger('dust3r_amd').error('d3r_aligner_set_image_range(all) failed with code %d after the sharded loop', rc)
    # every rank's own rows of the log-depth maps -> all ranks (sum with zeros elsewhere: exact)
    depth = net.im_depthmaps.data
    own = torch.zeros_like(depth)
    own[first:first + count] = depth[first:first + count]
    dist.all_reduce(own, op=dist.ReduceOp.SUM, group=group)
    depth.copy_(own)
    return loss


def _device_view(address, numel, dtype, device):
    """A torch tensor over `numel` elements of device memory the engine owns (no copy, no ownership)."""
    class _Span:
        pass
    span = _Span()
    itemsize = torch.empty((), dtype=dtype).element_size()
    span.__cuda_array_interface__ = {'shape': (numel,), 'typestr': {torch.float64: '<f8', torch.float32: '<f4'}[dtype], 'data': (int(address), False), 'version': 2,
                                     'strides': None}
    t = torch.as_tensor(span, device=device)
    assert t.data_ptr() == int(address) and t.numel() == numel and t.element_size() == itemsize
    return t


@torch.no_grad()
def clean_pointcloud_hip(im_confs, K, cams, depthmaps, all_pts3d, tol=0.001, bad_conf=0):
    """The reference's `clean_pointcloud` (base_opt.py:369-405: a point of image i that projects IN FRONT of image j's depthmap while
    being less confident than the pixel it lands on gets its confidence clipped to `bad_conf`; images visited in order, each seeing
    the already cleaned confidences of the earlier ones), computed by d3r_clean_pointcloud: n launches, one thread per pixel walking
    the other cameras, instead of n (n - 1) rounds of ~15 elementwise torch kernels."""
    _lib.require_device()
    n = len(im_confs)
    dev = im_confs[0].device
    shapes = [tuple(c.shape) for c in im_confs]
    maxA = max(h * w for h, w in shapes)

    def stack(ts, tail=()):
        out = torch.zeros((n, maxA) + tail, dtype=torch.float32, device=dev)
        for i, t in enumerate(ts):
            out[i, :t.numel() // max(1, int(np.prod(tail)))] = t.reshape((-1,) + tail).float()
        return out
    conf = stack(im_confs)
    depth = stack(depthmaps)
    pts = stack(all_pts3d, (3,))
    Kc = K.float().contiguous().reshape(n, 9)
    w2c = cams.float().contiguous().reshape(n, 16)
    hs = torch.tensor([h for h, w in shapes], dtype=torch.int32).to(dev)
    ws = torch.tensor([w for h, w in shapes], dtype=torch.int32).to(dev)
    with torch.cuda.device(dev):
        check(lib.d3r_clean_pointcloud(n, ptr(conf), ptr(depth), ptr(pts), ptr(Kc), ptr(w2c), ptr(hs), ptr(ws), maxA, float(tol), float(bad_conf),
                                       current_stream()), 'clean_pointcloud')
    return [conf[i, :h * w].view(h, w).to(im_confs[i].dtype) for i, (h, w) in enumerate(shapes)]
