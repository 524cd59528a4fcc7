"""Scene bootstrap: the one-shot initialisation of the global aligner, GPU-first.

Replaces what the reference does in `dust3r/cloud_opt/init_im_poses.py` (`init_minimum_spanning_tree`,
`init_from_known_poses`, `init_from_pts3d`, `minimum_spanning_tree`, `fast_pnp`) and what `PairViewer.__init__` does
(`pair_viewer.py:30-76`). The reference walks the spanning tree on the host and runs one roma / OpenCV call per step over
full-resolution clouds; here the walk is only a PLAN (which raw pairwise pointmap anchors every image, which registrations are
needed), every registration / focal fit / PnP pass over the big data is a batched HIP launch (csrc/bootstrap.hip), and the host
composes 4x4 similarities in fp64:

    pts3d[k]  ==  G_k applied to ONE raw map (side, e) of the network output         (never materialised)
    reg(x -> G y) == G o reg(x -> y)   for weighted Umeyama                           (all registrations independent)

The numbers are the reference's (same tree, same per-edge similarities, same Weiszfeld iterations) up to fp32 summation order;
PnP is this package's own solver (cloud_opt/pnp.py: OpenCV's RANSAC is not reproducible anyway).
"""
import ctypes as C
import math

import numpy as np
import scipy.sparse as sp
import torch

from .. import _lib
from .._lib import check, current_stream, lib, ptr
from . import pnp as pnp_host


# ---------------------------------------------------------------------------------------------------- device-side services
class PairMaps:
    """Addresses of the raw pairwise maps of a scene (`_stacked_pred_i/j`: (E, max_area, 3), `_conf_i/j`: (E, max_area))
    and the batched kernels that read them. A map is named (side, e): side 0 = view-1 prediction of edge e, 1 = view-2."""

    def __init__(self, scene):
        _lib.require_device()
        self.scene = scene
        self.dev = scene._stacked_pred_i.device
        if self.dev.type != 'cuda':
            raise _lib.D3RError('scene bootstrap runs on the GPU: build the aligner with device="cuda" (dust3r_amd has no CPU path)')
        self.preds = (scene._stacked_pred_i, scene._stacked_pred_j)
        self.confs = (scene._conf_i, scene._conf_j)
        self.max_area = scene.max_area
        self.edges = scene.edges
        self.imshapes = scene.imshapes

    def map_addr(self, side, e):
        t = self.preds[side]
        return t.data_ptr() + e * t.stride(0) * 4

    def conf_addr(self, side, e):
        t = self.confs[side]
        return t.data_ptr() + e * t.stride(0) * 4

    def npix(self, side, e):
        h, w = self.imshapes[self.edges[e][side]]
        return h * w

    def _i64(self, values):
        return torch.tensor(values, dtype=torch.int64).to(self.dev)

    def _i32(self, values):
        return torch.tensor(values, dtype=torch.int32).to(self.dev)

    # -- edge scores: mean(conf_i[e]) * mean(conf_j[e])  (commons.py:20-25)
    def edge_conf_means(self):
        out = []
        for side in (0, 1):
            t = self.confs[side]
            areas = {self.npix(side, e) for e in range(len(self.edges))}
            m = torch.empty(len(self.edges), dtype=torch.float32, device=self.dev)
            if len(areas) == 1:
                with torch.cuda.device(self.dev):
                    check(lib.d3r_row_means(ptr(t), t.shape[0], areas.pop(), t.stride(0), ptr(m), current_stream()), 'row_means')
            else:       # mixed image sizes: one launch per row length
                for a in areas:
                    rows = [e for e in range(len(self.edges)) if self.npix(side, e) == a]
                    for e in rows:
                        with torch.cuda.device(self.dev):
                            check(lib.d3r_row_means(C.c_void_p(self.conf_addr(side, e)), 1, a, t.stride(0), C.c_void_p(m.data_ptr() + 4 * e),
                                                    current_stream()), 'row_means')
            out.append(m)
        return out[0].double().cpu().numpy(), out[1].double().cpu().numpy()

    # -- batched weighted Umeyama moments; jobs: list of (src (side, e), tgt (side, e), weight (side, e)) -> (n, 17) float64
    def similarity_moments(self, jobs):
        if not jobs:
            return np.zeros((0, 17))
        n = len(jobs)
        src = self._i64([self.map_addr(*j[0]) for j in jobs])
        tgt = self._i64([self.map_addr(*j[1]) for j in jobs])
        wgt = self._i64([self.conf_addr(*j[2]) for j in jobs])
        npix = [self.npix(*j[0]) for j in jobs]
        assert all(self.npix(*j[1]) == a for j, a in zip(jobs, npix)), 'registration between clouds of different sizes'
        npix_d = self._i32(npix)
        out = torch.empty((n, 17), dtype=torch.float64, device=self.dev)
        with torch.cuda.device(self.dev):
            ws = torch.empty(int(lib.d3r_similarity_moments_workspace(n, max(npix))), dtype=torch.uint8, device=self.dev)
            check(lib.d3r_similarity_moments(n, ptr(src), ptr(tgt), ptr(wgt), ptr(npix_d), max(npix), ptr(ws), ptr(out), current_stream()),
                  'similarity_moments')
        return out.cpu().numpy()

    # -- Weiszfeld focal of the maps (side, e), principal point at the image centre (post_process.py:40-56)
    def weiszfeld_focals(self, maps, iterations=10):
        if not maps:
            return np.zeros(0)
        addr = self._i64([self.map_addr(*m) for m in maps])
        shapes = [self.imshapes[self.edges[e][side]] for side, e in maps]
        hs, ws = self._i32([h for h, w in shapes]), self._i32([w for h, w in shapes])
        out = torch.empty(len(maps), dtype=torch.float32, device=self.dev)
        with torch.cuda.device(self.dev):
            check(lib.d3r_weiszfeld_focals(len(maps), ptr(addr), ptr(hs), ptr(ws), iterations, ptr(out), current_stream()), 'weiszfeld_focals')
        return out.double().cpu().numpy()

    # -- (log-)depth of image k from its anchor map and the z row of (world->camera) o G_k, written into `out` (n, max_area)
    def anchor_depth(self, anchors, rows, out, take_log=True):
        n = len(anchors)
        addr = self._i64([self.map_addr(*a) for a in anchors])
        npix = self._i32([self.npix(*a) for a in anchors])
        rows_d = torch.tensor(np.asarray(rows, np.float32)).to(self.dev).contiguous()
        assert out.shape == (n, self.max_area) and out.is_contiguous() and out.dtype == torch.float32
        with torch.cuda.device(self.dev):
            check(lib.d3r_anchor_depth(n, ptr(addr), ptr(rows_d), ptr(npix), self.max_area, int(take_log), ptr(out), current_stream()), 'anchor_depth')
        return out


    # -- batched PnP (hypotheses on the host, consensus / refit / polish sums on the GPU)
    def solve_pnp(self, jobs, iterations=10):
        return solve_pnp_batch(self.dev, jobs, iterations=iterations)


# ---------------------------------------------------------------------------------------------------- host-side 4x4 algebra (fp64)
def similarity_from_moments(m):
    """(17,) weighted moments of source x and target y -> 4x4 similarity S with S x ~ y (weighted Umeyama), plus its scale."""
    W = m[0]
    mx, my = m[1:4] / W, m[4:7] / W
    cov = (m[7:16].reshape(3, 3) / W).T - np.outer(my, mx)          # sum w (y - my)(x - mx)^T / W
    var = m[16] / W - mx @ mx
    U, S, Vt = np.linalg.svd(cov)
    d = np.ones(3)
    d[2] = np.sign(np.linalg.det(U @ Vt)) or 1.0
    R = (U * d) @ Vt
    s = float((S * d).sum() / var)
    out = np.eye(4)
    out[:3, :3] = s * R
    out[:3, 3] = my - s * R @ mx
    return out


def similarities_from_moments(M):
    """`similarity_from_moments` for a stack (b, 17) -> (b, 4, 4) in one batched SVD (an initialisation at 100 views / 600 edges forms ~650 of them)."""
    M = np.asarray(M, np.float64).reshape(-1, 17)
    b = len(M)
    if b == 0:
        return np.zeros((0, 4, 4))
    W = M[:, 0:1]
    mx, my = M[:, 1:4] / W, M[:, 4:7] / W
    cov = np.swapaxes(M[:, 7:16].reshape(b, 3, 3) / W[:, :, None], 1, 2) - my[:, :, None] * mx[:, None, :]
    var = M[:, 16] / W[:, 0] - (mx * mx).sum(axis=1)
    U, S, Vt = np.linalg.svd(cov)
    d = np.ones((b, 3))
    sg = np.sign(np.linalg.det(U @ Vt))
    d[:, 2] = np.where(sg == 0, 1.0, sg)
    R = (U * d[:, None, :]) @ Vt
    s = (S * d).sum(axis=1) / var
    out = np.tile(np.eye(4), (b, 1, 1))
    out[:, :3, :3] = s[:, None, None] * R
    out[:, :3, 3] = my - s[:, None] * np.einsum('bij,bj->bi', R, mx)
    return out


def split_similarity(G):
    """4x4 [sR | t] -> (s, R, t)."""
    A = G[:3, :3]
    s = float(np.cbrt(np.linalg.det(A)))
    return s, A / s, G[:3, 3].copy()


def rigid_part(G):
    s, R, t = split_similarity(G)
    out = np.eye(4)
    out[:3, :3], out[:3, 3] = R, t
    return out


def rotmat_to_quat_xyzw(R):
    m = np.asarray(R, np.float64)
    d = np.array([1 + m[0, 0] - m[1, 1] - m[2, 2], 1 - m[0, 0] + m[1, 1] - m[2, 2], 1 - m[0, 0] - m[1, 1] + m[2, 2], 1 + m[0, 0] + m[1, 1] + m[2, 2]])
    c = int(d.argmax())
    q = [np.array([d[0], m[1, 0] + m[0, 1], m[0, 2] + m[2, 0], m[2, 1] - m[1, 2]]),
         np.array([m[1, 0] + m[0, 1], d[1], m[2, 1] + m[1, 2], m[0, 2] - m[2, 0]]),
         np.array([m[0, 2] + m[2, 0], m[2, 1] + m[1, 2], d[2], m[1, 0] - m[0, 1]]),
         np.array([m[2, 1] - m[1, 2], m[0, 2] - m[2, 0], m[1, 0] - m[0, 1], d[3]])][c]
    return q / np.linalg.norm(q)


def pose_params(R, T, scale=None):
    """(quat XYZW, signed-log translation[, log scale]) -- the aligner's pose parameterisation (base_opt.py:157-176)."""
    T = np.asarray(T, np.float64) / (scale if scale is not None else 1.0)
    p = np.concatenate((rotmat_to_quat_xyzw(R), np.sign(T) * np.log1p(np.abs(T))))
    with np.errstate(divide='ignore', invalid='ignore'):
        return p if scale is None else np.concatenate((p, [float(np.log(np.float64(scale)))]))


def pose_params_batch(M, with_scale):
    """`pose_params(*split_similarity(M)[1:], scale)` for a stack of 4x4 [sR | t] (b, 4, 4) -> (b, 8) (with_scale: quaternion XYZW, signed-log of t / s, log s)
    or (b, 7) (rigid: s is not divided out and not stored)."""
    M = np.asarray(M, np.float64)
    A = M[:, :3, :3]
    sc = np.cbrt(np.linalg.det(A))
    m = A / sc[:, None, None] if with_scale else A
    T = M[:, :3, 3] / sc[:, None] if with_scale else M[:, :3, 3]
    d = np.stack((1 + m[:, 0, 0] - m[:, 1, 1] - m[:, 2, 2], 1 - m[:, 0, 0] + m[:, 1, 1] - m[:, 2, 2], 1 - m[:, 0, 0] - m[:, 1, 1] + m[:, 2, 2],
                  1 + m[:, 0, 0] + m[:, 1, 1] + m[:, 2, 2]), axis=1)
    c = d.argmax(axis=1)
    a, bq, cq = m[:, 1, 0] + m[:, 0, 1], m[:, 0, 2] + m[:, 2, 0], m[:, 2, 1] + m[:, 1, 2]
    x, y, z = m[:, 2, 1] - m[:, 1, 2], m[:, 0, 2] - m[:, 2, 0], m[:, 1, 0] - m[:, 0, 1]
    cands = np.stack((np.stack((d[:, 0], a, bq, x), 1), np.stack((a, d[:, 1], cq, y), 1), np.stack((bq, cq, d[:, 2], z), 1), np.stack((x, y, z, d[:, 3]), 1)), axis=1)
    q = cands[np.arange(len(M)), c]
    q = q / np.linalg.norm(q, axis=1, keepdims=True)
    out = np.concatenate((q, np.sign(T) * np.log1p(np.abs(T))), axis=1)
    if with_scale:
        with np.errstate(divide='ignore', invalid='ignore'):
            out = np.concatenate((out, np.log(sc)[:, None]), axis=1)
    return out


def align_pose_sets(src, dst):
    """Similarity that maps the camera centres (+ a point a little down each optical axis) of `src` poses onto `dst` poses
    (init_im_poses.py:308-316), as a 4x4 and its (s, R, t)."""
    src, dst = np.asarray(src, np.float64), np.asarray(dst, np.float64)

    def centres_and_axes(P):
        c = P[:, :3, 3]
        d = np.linalg.norm(c[:, None] - c[None], axis=-1)[np.triu_indices(len(c), 1)]
        eps = np.median(d) / 100                     # geometry.py:364-366 get_med_dist_between_poses: median of the CONDENSED pairwise distances (scipy pdist)
        return np.concatenate((c, c + eps * P[:, :3, 2]))
    x, y = centres_and_axes(src), centres_and_axes(dst)
    m = np.zeros(17)
    m[0] = len(x)
    m[1:4], m[4:7] = x.sum(0), y.sum(0)
    m[7:16] = (x[:, :, None] * y[:, None, :]).sum(0).ravel()
    m[16] = (x * x).sum()
    return similarity_from_moments(m)


# ---------------------------------------------------------------------------------------------------- the spanning-tree plan
class TreePlan:
    """The reference's walk over the maximum-confidence spanning tree (init_im_poses.py:127-183), recorded instead of executed:
    anchor[k]   raw map (side, e) whose transform IS image k's world cloud
    parent[k]   (parent image or None, job index or None): G_k = G_parent o S_job
    pose_job[k] job whose composed similarity gives image k's camera pose (k led a tree edge), 'identity' for the root, else None
    focal_map[k] map (0, e) the reference fits image k's focal on (including its stale-edge quirk), or None"""

    def __init__(self, n_imgs):
        self.anchor = [None] * n_imgs
        self.parent = [None] * n_imgs
        self.pose_job = [None] * n_imgs
        self.focal_map = [None] * n_imgs
        self.jobs = []            # (src map, tgt map, weight map)
        self.order = []           # images in placement order
        self.tree_edges = []


def plan_spanning_tree(n_imgs, edges, mean_i, mean_j):
    edge_index = {ij: e for e, ij in enumerate(edges)}
    graph = sp.dok_array((n_imgs, n_imgs))
    for e, (i, j) in enumerate(edges):
        graph[i, j] = float(mean_i[e] * mean_j[e])
    graph = -graph
    tree = sp.csgraph.minimum_spanning_tree(graph).tocoo()
    pending = sorted(zip(-tree.data, tree.row.tolist(), tree.col.tolist()))
    plan = TreePlan(n_imgs)

    _, i, j = pending.pop()                          # strongest edge: camera i is the world frame
    e = edge_index[(i, j)]
    plan.anchor[i], plan.anchor[j] = (0, e), (1, e)
    plan.parent[i], plan.parent[j] = (None, None), (None, None)
    plan.pose_job[i] = 'identity'
    plan.focal_map[i] = (0, e)
    plan.order += [i, j]
    plan.tree_edges.append((i, j))
    placed = {i, j}
    last_e = e
    while pending:
        score, i, j = pending.pop()
        if plan.focal_map[i] is None:
            plan.focal_map[i] = (0, last_e)         # the reference fits on pred_i of the PREVIOUS edge here (init_im_poses.py:152-153)
        if i in placed or j in placed:
            e = edge_index[(i, j)]
            old, new, side = (i, j, 0) if i in placed else (j, i, 1)
            # register this edge's map of the placed image onto that image's anchor, weighted by the edge's confidence on that side
            plan.jobs.append(((side, e), plan.anchor[old], (side, e)))
            job = len(plan.jobs) - 1
            plan.anchor[new] = (1 - side, e)
            plan.parent[new] = (old, job)
            if plan.pose_job[i] is None:             # either way the edge frame is camera i's frame
                plan.pose_job[i] = (old, job)
            placed.add(new)
            plan.order.append(new)
            plan.tree_edges.append((i, j))
            last_e = e
        else:
            pending.insert(0, (score, i, j))        # neither end placed yet
    # focals still missing: best-scoring edge whose FIRST image it is (init_im_poses.py:188-192)
    keys = np.array(list(graph.keys()))
    for i, j in keys[np.argsort(list(graph.values()))].tolist():
        if plan.focal_map[i] is None:
            plan.focal_map[i] = (0, edge_index[(i, j)])
    return plan


# ---------------------------------------------------------------------------------------------------- batched PnP on the GPU
class _PnpJobRec(C.Structure):
    _fields_ = [('map', C.c_void_p), ('conf', C.c_void_p), ('G', C.c_float * 12), ('f', C.c_float), ('ppx', C.c_float), ('ppy', C.c_float),
                ('thr', C.c_float), ('H', C.c_int), ('W', C.c_int)]


def solve_pnp_batch(dev, jobs, iterations=10, reproj_err=5.0, seed=0, refine_iters=10):
    """jobs: list of dict(map=addr of (H,W,3), conf=addr of (H,W), G=3x4 (world = G . map point), f, pp=(x, y), thr, H, W,
    points=torch (H,W,3) view, confs=torch (H,W) view). Returns per job (success, world->cam 4x4, inlier count).
    Hypotheses: 6-point DLT on random masked points (host, tiny); consensus scoring and the Gauss-Newton sums of the polish: GPU, all
    jobs in the same launches."""
    assert lib.d3r_pnp_job_bytes() == C.sizeof(_PnpJobRec)
    n = len(jobs)
    if n == 0:
        return []
    maxh = int(lib.d3r_pnp_max_hypotheses())
    # `iterations` is the reference's RANSAC budget for OpenCV's SQPnP minimal solver; a linear DLT hypothesis is cheaper and noisier,
    # and scoring costs nothing extra on the GPU (one pass evaluates every hypothesis): always use the kernel's full set, each from
    # NSAMPLE points (over-determined: pointmap noise averages out; DUSt3R pointmaps have few gross outliers)
    nh = maxh
    NSAMPLE = 12
    nv = int(lib.d3r_pnp_sum_count())
    recs = (_PnpJobRec * n)()
    for r, j in zip(recs, jobs):
        r.map, r.conf = j['map'], j['conf']
        for k, v in enumerate(np.asarray(j['G'], np.float32).reshape(12)):
            r.G[k] = float(v)
        r.f, r.ppx, r.ppy, r.thr, r.H, r.W = float(j['f']), float(j['pp'][0]), float(j['pp'][1]), float(j['thr']), int(j['H']), int(j['W'])
    recs_d = torch.frombuffer(bytearray(bytes(recs)), dtype=torch.uint8).to(dev)
    rng = np.random.RandomState(seed)

    # ---- hypotheses from minimal samples (the samples are a few hundred points: gathered to the host)
    # Every job's candidate pixels in ONE device gather and one copy back, every hypothesis' DLT in one batched call (pnp.dlt_pose_batch): the
    # per-job loop of rounds 2-4 paid two synchronising copies per job and 62 us of numpy call overhead per hypothesis -- 0.08 of the 0.13 s of
    # init='mst' at 100 views.
    hyp = np.zeros((n, maxh, 12), np.float32)
    valid = np.zeros((n, maxh), bool)
    cand = np.stack([rng.randint(0, j['H'] * j['W'], size=nh * 48) for j in jobs])       # drawn for every job: the stream does not depend on failures
    live = [a for a, j in enumerate(jobs) if np.isfinite(j['f']) and j['f'] > 0]     # degenerate intrinsics: no hypothesis, the job fails
    if live:
        cand_d = torch.from_numpy(cand).to(dev)
        pts_d = torch.stack([jobs[a]['points'].reshape(-1, 3)[cand_d[a]] for a in live]).double()
        ok_d = torch.stack([jobs[a]['confs'].reshape(-1)[cand_d[a]] > jobs[a]['thr'] for a in live])
        pts, ok = pts_d.cpu().numpy(), ok_d.cpu().numpy()
        G = np.stack([np.asarray(jobs[a]['G'], np.float64).reshape(3, 4) for a in live])
        world = np.einsum('apk,aik->api', pts, G[:, :, :3]) + G[:, None, :, 3]
        Wl = np.array([jobs[a]['W'] for a in live])[:, None]
        fl = np.array([jobs[a]['f'] for a in live], np.float64)[:, None]
        ppl = np.array([jobs[a]['pp'] for a in live], np.float64)
        cl = cand[live]
        xn = np.stack((((cl % Wl) - ppl[:, 0:1]) / fl, ((cl // Wl) - ppl[:, 1:2]) / fl), axis=2)
        # hypothesis h of a job: the first NSAMPLE confident candidates of its window of 48
        okw = ok.reshape(len(live) * nh, 48)
        sel = okw & (np.cumsum(okw, axis=1) <= NSAMPLE)
        first = np.argsort(~sel, axis=1, kind='stable')[:, :NSAMPLE]           # the selected candidates first, in their order: NSAMPLE columns instead of 48
        R, T, good = pnp_host.dlt_pose_batch(np.take_along_axis(world.reshape(len(live) * nh, 48, 3), first[:, :, None], axis=1),
                                             np.take_along_axis(xn.reshape(len(live) * nh, 48, 2), first[:, :, None], axis=1), np.take_along_axis(sel, first, axis=1))
        sol = np.concatenate((R, T[:, :, None]), axis=2).reshape(len(live), nh, 12)
        good = good.reshape(len(live), nh)
        hyp[live, :nh] = np.where(good[:, :, None], sol, 0.0)
        valid[live, :nh] = good
    hyp_d = torch.from_numpy(hyp).to(dev)
    counts = torch.zeros((n, maxh), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        check(lib.d3r_pnp_score(n, ptr(recs_d), ptr(hyp_d), nh, float(reproj_err), ptr(counts), current_stream()), 'pnp_score')
    counts = counts.cpu().numpy()
    counts[~valid] = 0
    best = counts[:, :nh].argmax(axis=1)
    poses = np.stack([hyp[a, best[a]].reshape(3, 4) for a in range(n)]).astype(np.float64)
    best_count = counts[np.arange(n), best]

    ws = torch.empty(int(lib.d3r_pnp_workspace(n)), dtype=torch.uint8, device=dev)
    sums = torch.empty((n, nv), dtype=torch.float64, device=dev)

    def gn_sums(P):
        P_d = torch.from_numpy(np.ascontiguousarray(P.reshape(n, 12), np.float32)).to(dev)
        with torch.cuda.device(dev):
            check(lib.d3r_pnp_sums(n, ptr(recs_d), ptr(P_d), float(reproj_err), ptr(ws), ptr(sums), current_stream()), 'pnp_sums')
        return sums.cpu().numpy()

    # polish: Gauss-Newton on the reprojection error of the consensus set (re-selected at every step). The converged pose replaces the
    # minimal-sample hypothesis unless it lost more than 5 % of the consensus (a 6-point hypothesis that is a few 1e-3 off can hold a
    # handful of stray points inside the 5-pixel band that the least-squares pose drops: count alone must not veto the polish).
    ju = np.triu_indices(6)
    cand = poses.copy()
    hyp_count = best_count.astype(np.float64)
    eye6 = 1e-9 * np.eye(6)
    for _ in range(refine_iters):
        g = gn_sums(cand)
        use = (g[:, 28] >= 6) & np.isfinite(g).all(axis=1)
        Hm = np.zeros((n, 6, 6))
        Hm[:, ju[0], ju[1]] = g[:, 0:21]
        Hm = Hm + np.swapaxes(Hm, 1, 2) - Hm * np.eye(6) + eye6
        Hm[~use] = np.eye(6)
        try:
            d = np.linalg.solve(Hm, -g[:, 21:27, None])[:, :, 0]
        except np.linalg.LinAlgError:                                   # a singular system somewhere in the stack: job by job, skipping those
            d = np.zeros((n, 6))
            for a in np.nonzero(use)[0]:
                try:
                    d[a] = np.linalg.solve(Hm[a], -g[a, 21:27])
                except np.linalg.LinAlgError:
                    use[a] = False
        d[~use] = 0.0
        for a in np.nonzero(use)[0]:
            cand[a, :, :3] = pnp_host.rodrigues_to_rotmat(d[a, :3]) @ cand[a, :, :3]   # R <- exp([w]x) R about the camera origin,
        cand[:, :, 3] += d[:, 3:]                                                       # t <- t + dt (the kernel's Jacobian convention)
        if float(np.linalg.norm(d, axis=1).max(initial=0.0)) < 1e-9:
            break
    g = gn_sums(cand)
    best_count = best_count.astype(np.float64)
    for a in range(n):
        if g[a, 28] >= 0.95 * hyp_count[a]:
            poses[a], best_count[a] = cand[a], g[a, 28]
    out = []
    for a in range(n):
        M = np.eye(4)
        M[:3] = poses[a]
        out.append((bool(best_count[a] >= 6), M, int(best_count[a])))
    return out


# ---------------------------------------------------------------------------------------------------- the two initialisations
@torch.no_grad()
def bootstrap_from_spanning_tree(scene, niter_PnP=10, maps=None):
    """`init='mst'`: world clouds along the maximum-confidence spanning tree, focals, camera poses, then every pairwise pose,
    the scale normalisation and the per-image depth / pose / focal parameters (init_im_poses.py:67-123,127-209).
    `maps`: the device-side services (default: PairMaps(scene), the HIP kernels)."""
    maps = PairMaps(scene) if maps is None else maps
    n, edges = scene.n_imgs, scene.edges
    mean_i, mean_j = maps.edge_conf_means()
    plan = plan_spanning_tree(n, edges, mean_i, mean_j)
    if scene.verbose:
        print(f' init tree: {len(plan.tree_edges)} edges, root {plan.order[0]}')

    # every registration of the initialisation in one launch: tree edges, then (pred_i[e] -> anchor of image i) for the pairwise poses
    jobs = list(plan.jobs)
    pw_job = []
    for e, (i, j) in enumerate(edges):
        if plan.anchor[i] == (0, e):
            pw_job.append(None)                     # the cloud IS that map: identity
        else:
            jobs.append(((0, e), plan.anchor[i], (0, e)))
            pw_job.append(len(jobs) - 1)
    S = list(similarities_from_moments(maps.similarity_moments(jobs)))

    G = [None] * n
    for k in plan.order:
        par, job = plan.parent[k]
        G[k] = np.eye(4) if par is None else G[par] @ S[job]

    has_poses = scene.has_im_poses
    focals, poses = [None] * n, [None] * n
    if has_poses:
        fmaps = [(k, m) for k, m in enumerate(plan.focal_map) if m is not None]
        for (k, _), f in zip(fmaps, maps.weiszfeld_focals([m for _, m in fmaps])):
            focals[k] = float(f)
        for k in range(n):
            pj = plan.pose_job[k]
            if pj == 'identity':
                poses[k] = np.eye(4)
            elif pj is not None:
                poses[k] = rigid_part(G[pj[0]] @ S[pj[1]])
        need = [k for k in range(n) if poses[k] is None]
        if need:
            pjobs, owner = [], []
            confident = torch.stack([(scene.im_conf[k] > scene.min_conf_thr).sum() for k in need]).cpu().tolist()      # one copy back for all images
            for k, n_conf in zip(need, confident):
                side, e = plan.anchor[k]
                H, W = scene.imshapes[k]
                if n_conf < 4:
                    continue
                conf_t = scene.im_conf[k].contiguous()
                sweep = [focals[k]] if focals[k] is not None else list(np.geomspace(max(W, H) / 2, max(W, H) * 3, 21))
                for f in sweep:
                    pjobs.append(dict(map=maps.map_addr(side, e), conf=conf_t.data_ptr(), G=G[k][:3], f=f, pp=(W / 2, H / 2), thr=scene.min_conf_thr,
                                      H=H, W=W, points=maps.preds[side][e][:H * W], confs=conf_t, keep=conf_t))
                    owner.append((k, f))
            best = {}
            for (k, f), (ok, w2c, cnt) in zip(owner, maps.solve_pnp(pjobs, iterations=niter_PnP)):
                if ok and cnt > best.get(k, (0,))[0]:
                    best[k] = (cnt, f, w2c)
            for k, (cnt, f, w2c) in best.items():
                focals[k], poses[k] = float(f), np.linalg.inv(w2c)
        for k in range(n):
            if poses[k] is None:
                poses[k] = np.eye(4)
    _commit(scene, maps, plan.anchor, G, S, pw_job, focals, poses)


def _commit(scene, maps, anchor, G, S, pw_job, focals, poses):
    """World clouds (anchor, G), pairwise registrations and per-image focals / poses -> the aligner's parameters
    (init_im_poses.py:83-123)."""
    n, edges = scene.n_imgs, scene.edges
    has_poses = scene.has_im_poses
    poses = None if not has_poses else np.stack(poses)
    if has_poses:
        known = np.array([not scene.im_poses.requires_grad] * n)
        nkp = int(known.sum())
        if nkp == 1:
            raise NotImplementedError('Would be simpler to just align everything afterwards on the single known pose')
        if nkp > 1:                                 # global similarity onto the preset poses
            target = scene.get_im_poses().detach().double().cpu().numpy()
            trf = align_pose_sets(poses[known], target[known])
            s = split_similarity(trf)[0]
            poses = trf @ poses
            poses[:, :3, :3] /= s
            G = [trf @ g for g in G]
    # pairwise poses: cloud of image i as seen from edge e's frame
    pw = pose_params_batch(np.stack([G[i] if pw_job[e] is None else G[i] @ S[pw_job[e]] for e, (i, j) in enumerate(edges)]), True).astype(np.float32)
    if scene.pw_poses.requires_grad:
        scene.pw_poses.data.copy_(torch.from_numpy(pw).to(scene.pw_poses.device))
    s_factor = float(scene.get_pw_norm_scale_factor())
    G = [np.diag([s_factor, s_factor, s_factor, 1.0]) @ g for g in G]
    if not has_poses:
        return
    poses[:, :3, 3] *= s_factor
    if scene.im_depthmaps.requires_grad:
        rows = [(np.linalg.inv(poses[k]) @ G[k])[2] for k in range(n)]
        maps.anchor_depth([anchor[k] for k in range(n)], rows, scene.im_depthmaps.data)
    if scene.im_poses.requires_grad:
        scene.im_poses.data.copy_(torch.from_numpy(pose_params_batch(poses, False).astype(np.float32)).to(scene.im_poses.device))
    if scene.im_focals.requires_grad:
        vals = scene.im_focals.data.clone()
        for k, f in enumerate(focals):
            if f is not None:
                with np.errstate(divide='ignore', invalid='ignore'):     # a degenerate focal gives -inf / nan like the reference's np.log, not an exception
                    vals[k] = scene.focal_break * float(np.log(np.float64(f)))
        scene.im_focals.data.copy_(vals)
    if scene.verbose:
        print(' init loss =', float(scene()))


@torch.no_grad()
def bootstrap_from_known_poses(scene, niter_PnP=10, min_conf_thr=3, maps=None):
    """`init='known_poses'` (init_im_poses.py:24-63): every camera pose and focal is preset; each edge gets its pairwise pose from a
    PnP of its view-2 cloud and a two-camera alignment onto the preset poses; each image takes the depth of its most confident edge."""
    n, edges = scene.n_imgs, scene.edges
    assert not scene.im_poses.requires_grad, 'not all poses are known'
    assert bool(scene.get_known_focal_mask().all()), 'not all focals are known'
    maps = PairMaps(scene) if maps is None else maps
    known = scene.get_im_poses().detach().double().cpu().numpy()
    focals = scene.get_focals().detach().double().cpu().numpy().reshape(n, -1).mean(axis=1)
    pps = scene.get_principal_points().detach().double().cpu().numpy()
    pjobs = []
    for e, (i, j) in enumerate(edges):
        H, W = scene.imshapes[j]
        # the reference masks with conf_i > min(thr, conf_i.min() - 0.1): true everywhere by construction
        pjobs.append(dict(map=maps.map_addr(1, e), conf=maps.conf_addr(0, e), G=np.eye(4)[:3], f=focals[i], pp=pps[i], thr=-np.inf, H=H, W=W,
                          points=maps.preds[1][e][:H * W], confs=maps.confs[0][e][:H * W]))
    sols = maps.solve_pnp(pjobs, iterations=niter_PnP)
    mean_i, _ = maps.edge_conf_means()
    pw = np.zeros((len(edges), 8), np.float32)
    best = {}
    for e, (i, j) in enumerate(edges):
        ok, w2c, _ = sols[e]
        assert ok, f'PnP failed on edge {i}_{j}'
        trf = align_pose_sets(np.stack((np.eye(4), np.linalg.inv(w2c))), known[[i, j]])
        s, R, T = split_similarity(trf)
        pw[e] = pose_params(R, T, scale=s)
        if mean_i[e] > best.get(i, (0,))[0]:
            best[i] = (mean_i[e], e, s)
    scene.pw_poses.data.copy_(torch.from_numpy(pw).to(scene.pw_poses.device))
    rows = [np.array([0, 0, best[k][2], 0.0]) for k in range(n)]                   # depth = z of pred_i[best edge] * scale
    maps.anchor_depth([(0, best[k][1]) for k in range(n)], rows, scene.im_depthmaps.data)
