"""gloo tests (CPU, world size 2 and 4) of the pair-sharded inference path (dust3r_amd/parallel.py): the sharded result must equal the
single-process `inference()` result bit for bit, including ragged shards (odd pair counts) and BASELINE configs[2]'s 190 pairs."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class StandInModel:
    """Deterministic per-pair 'network' (pure function of the two images) so that the test exercises the
    sharding / packing / all-gather plumbing without a GPU. NOT a compute fallback of the product."""

    def __call__(self, view1, view2):
        a, b = view1['img'], view2['img']
        B, _, H, W = a.shape
        pts1 = (a * 2 + b).permute(0, 2, 3, 1).contiguous()
        pts2 = (a - b * 3).permute(0, 2, 3, 1).contiguous()
        conf1 = 1 + (a * b).sum(1).abs()
        conf2 = 1 + (a + b).sum(1).abs()
        return dict(pts3d=pts1, conf=conf1), dict(pts3d_in_other_view=pts2, conf=conf2)


class StandInEncodeOnce(StandInModel):
    """The same function split the way the engine's encode-once API splits it (encode_images / decode_pairs), counting encoder calls."""
    _engine = object()

    def __init__(self):
        self.encoded = 0

    def encode_images(self, img):
        self.encoded += img.shape[0]
        return img.reshape(img.shape[0], -1).clone()

    def decode_pairs(self, feat, H, W, packed_out=None):
        B = feat.shape[0] // 2
        a, b = feat[:B].reshape(B, 3, H, W), feat[B:].reshape(B, 3, H, W)
        return StandInModel.__call__(self, dict(img=a), dict(img=b))


class StandInMixed:
    """Stand-in for pairs whose two views differ in size: view 1's outputs at view 1's size, view 2's at view 2's, each depending on both images."""

    def __call__(self, view1, view2):
        a, b = view1['img'], view2['img']
        ma, mb = a.mean(dim=(2, 3), keepdim=True), b.mean(dim=(2, 3), keepdim=True)
        pts1 = (a * 2 + mb).permute(0, 2, 3, 1).contiguous()
        pts2 = (b * 3 - ma).permute(0, 2, 3, 1).contiguous()
        return dict(pts3d=pts1, conf=1 + (a * mb).sum(1).abs()), dict(pts3d_in_other_view=pts2, conf=1 + (b + ma).sum(1).abs())


def _mixed_pairs():
    """7 images of three sizes -> 42 ordered pairs of mixed (view 1, view 2) sizes, the list dust3r/inference.py:60-68 runs one by one"""
    g = torch.Generator().manual_seed(4)
    shapes = [(16, 32), (32, 16), (16, 32), (16, 16), (32, 16), (16, 32), (16, 32)]
    imgs = [dict(img=torch.rand((1, 3, h, w), generator=g) * 2 - 1, true_shape=torch.tensor([[h, w]], dtype=torch.int32), idx=k, instance=str(k))
            for k, (h, w) in enumerate(shapes)]
    return [(imgs[i], imgs[j]) for i in range(len(imgs)) for j in range(len(imgs)) if i != j]


def _pairs(n_views, H, W):
    sys.path.insert(0, ROOT)
    from dust3r_amd.image_pairs import make_pairs
    from dust3r_amd.synthetic import synthetic_image_list
    return make_pairs(synthetic_image_list(n_views, H, W, seed=3), 'complete', None, symmetrize=False)


def _worker(rank, world, port, n_views, outdir, H=16, W=32, mode='plain'):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from dust3r_amd.parallel import inference_sharded
        if mode == 'mixed':
            out = inference_sharded(_mixed_pairs(), StandInMixed(), 'cpu', batch_size=3)
            torch.save((rank, out), os.path.join(outdir, f'rank{rank}.pt'))
            dist.barrier()
            return
        model = StandInEncodeOnce() if mode == 'encode_once' else StandInModel()
        from dust3r_amd.image_pairs import make_pairs
        from dust3r_amd.synthetic import synthetic_image_list
        pairs = make_pairs(synthetic_image_list(n_views, H, W, seed=3), 'complete', None, symmetrize=True) if mode == 'encode_once' else _pairs(n_views, H, W)
        out = inference_sharded(pairs, model, 'cpu', batch_size=2)
        if mode == 'encode_once':
            # a rank encodes the distinct images of ITS shard once: fewer encoder passes than its 2 x (pairs in the shard) view slots
            from dust3r_amd.parallel import shard_plan
            plan = shard_plan(pairs, world, encode_once=True)
            mine = plan.shard(rank)
            touched = {int(v['idx']) for k in mine for v in pairs[k]}
            assert model.encoded == len(touched) == plan.images[rank] < 2 * len(mine), (model.encoded, len(touched), len(mine))
        torch.save((rank, out['pred1']['pts3d'], out['pred1']['conf'], out['pred2']['pts3d_in_other_view'], out['pred2']['conf'],
                    out['view1']['idx'], out['view2']['idx']), os.path.join(outdir, f'rank{rank}.pt'))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


# 6 pairs on 2 ranks (even shards), 3 pairs on 2 ranks (ragged: 2 + 1 + padding), and BASELINE configs[2]: 20 views -> 190 pairs on 4 ranks
# (48 + 48 + 48 + 46 with two padding slots)
@pytest.mark.parametrize('n_views,world,H,W', [(4, 2, 16, 32), (3, 2, 16, 32), (20, 4, 8, 16)])
def test_sharded_inference_equals_single_process(n_views, world, H, W, tmp_path):
    from dust3r_amd.inference import inference
    ref = inference(_pairs(n_views, H, W), StandInModel(), 'cpu', batch_size=2, verbose=False)
    assert len(ref['view1']['idx']) == n_views * (n_views - 1) // 2
    ctx = mp.get_context('spawn')
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_views, str(tmp_path), H, W)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=240)
        assert p.exitcode == 0
    got = [torch.load(os.path.join(str(tmp_path), f'rank{r}.pt')) for r in range(world)]
    for rank, pts1, conf1, pts2, conf2, idx1, idx2 in got:
        assert torch.equal(pts1, ref['pred1']['pts3d']) and torch.equal(conf1, ref['pred1']['conf'])
        assert torch.equal(pts2, ref['pred2']['pts3d_in_other_view']) and torch.equal(conf2, ref['pred2']['conf'])
        assert idx1 == ref['view1']['idx'] and idx2 == ref['view2']['idx']


def _spawn(world, tmp_path, *args):
    ctx = mp.get_context('spawn')
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port) + args) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=240)
        assert p.exitcode == 0
    return [torch.load(os.path.join(str(tmp_path), f'rank{r}.pt'), weights_only=False) for r in range(world)]


def test_sharded_inference_encodes_each_image_of_a_shard_once(tmp_path):
    """The encode-once route of inference_sharded (what the engine takes on the GPU): 5 views, complete symmetrised graph = 20 pairs on
    2 ranks; every rank encodes only the images its shard touches, once; result bit-identical to the single-process pair-by-pair call."""
    sys.path.insert(0, ROOT)
    from dust3r_amd.image_pairs import make_pairs
    from dust3r_amd.inference import inference
    from dust3r_amd.synthetic import synthetic_image_list
    pairs = make_pairs(synthetic_image_list(5, 16, 32, seed=3), 'complete', None, symmetrize=True)
    ref = inference(pairs, StandInModel(), 'cpu', batch_size=2, verbose=False)
    for rank, pts1, conf1, pts2, conf2, idx1, idx2 in _spawn(2, tmp_path, 5, str(tmp_path), 16, 32, 'encode_once'):
        assert torch.equal(pts1, ref['pred1']['pts3d']) and torch.equal(conf1, ref['pred1']['conf'])
        assert torch.equal(pts2, ref['pred2']['pts3d_in_other_view']) and torch.equal(conf2, ref['pred2']['conf'])
        assert idx1 == ref['view1']['idx'] and idx2 == ref['view2']['idx']


@pytest.mark.parametrize('world', [2, 3])
def test_sharded_inference_with_mixed_image_sizes(world, tmp_path):
    """Pairs of several image sizes shard too: grouped by shape inside each shard, one flat padded payload, ONE all-gather; the result
    has the list-per-pair structure `inference()` returns for such a list (dust3r/inference.py:60-72), bit-identical."""
    from dust3r_amd.inference import inference
    pairs = _mixed_pairs()
    ref = inference(pairs, StandInMixed(), 'cpu', batch_size=1, verbose=False)
    assert isinstance(ref['pred1']['pts3d'], list) and len(ref['pred1']['pts3d']) == len(pairs) == 42
    for rank, out in _spawn(world, tmp_path, 0, str(tmp_path), 0, 0, 'mixed'):
        assert out['loss'] is None or out['loss'] == [None] * len(pairs) or out['loss'] == ref['loss']
        for view, keys in (('pred1', ('pts3d', 'conf')), ('pred2', ('pts3d_in_other_view', 'conf')), ('view1', ('img', 'true_shape')), ('view2', ('img', 'true_shape'))):
            for k in keys:
                a, b = ref[view][k], out[view][k]
                assert isinstance(b, list) and len(a) == len(b) and all(x.shape == y.shape and torch.equal(x, y) for x, y in zip(a, b)), (view, k)
        assert out['view1']['idx'] == ref['view1']['idx'] and out['view2']['instance'] == ref['view2']['instance']


def test_shard_bounds_cover_everything_once():
    from dust3r_amd.parallel import shard_bounds
    for n in (0, 1, 7, 8, 9, 190, 600):
        for world in (1, 2, 4, 8):
            seen = []
            for r in range(world):
                lo, hi, per = shard_bounds(n, r, world)
                assert hi - lo <= per
                seen += list(range(lo, hi))
            assert seen == list(range(n))


def _index_views(n, area=None):
    return [dict(idx=i, instance=str(i)) for i in range(n)]


@pytest.mark.parametrize('n_views,graph,sym', [(20, 'complete', False), (20, 'complete', True), (100, 'swin-3', True), (100, 'logwin-3', True), (7, 'oneref-2', True), (2, 'complete', False)])
def test_shard_plan_covers_every_pair_once_and_beats_contiguous_slices(n_views, graph, sym):
    """shard_plan (round 5): every pair on exactly one rank, `source` maps the gathered rows back to the caller's order, the plan is
    never worse than the plain contiguous ceil(P / N) slices under its own cost model (encoder pass per distinct image + decoder/head
    pass per pair) -- and for the windowed graphs, whose pair list comes out of a set in hash order, much better."""
    from dust3r_amd.image_pairs import make_pairs
    from dust3r_amd.parallel import DEC_COST_PER_PAIR, ENC_COST_PER_IMAGE, shard_bounds, shard_plan
    pairs = make_pairs(_index_views(n_views), graph, None, symmetrize=sym)
    P = len(pairs)
    for world in (1, 2, 3, 4, 8):
        plan = shard_plan(pairs, world, encode_once=True)
        assert sorted(k for r in range(world) for k in plan.shard(r)) == list(range(P))
        assert len(plan.bounds) == world and plan.per == max(plan.counts)
        rows = torch.full((world * max(plan.per, 1),), -1, dtype=torch.long)      # what the all-gather delivers: rank r's rows at r * per
        for r in range(world):
            mine = plan.shard(r)
            rows[r * plan.per:r * plan.per + len(mine)] = torch.tensor(mine, dtype=torch.long)
            assert plan.images[r] == len({int(v['idx']) for k in mine for v in pairs[k]})
            assert abs(plan.cost[r] - (ENC_COST_PER_IMAGE * plan.images[r] + DEC_COST_PER_PAIR * len(mine))) < 1e-6
        assert rows.index_select(0, plan.source).tolist() == list(range(P))
        contiguous = []
        for r in range(world):
            lo, hi, _ = shard_bounds(P, r, world)
            contiguous.append(ENC_COST_PER_IMAGE * len({int(v['idx']) for p in pairs[lo:hi] for v in p}) + DEC_COST_PER_PAIR * (hi - lo))
        assert max(plan.cost) <= max(contiguous) * (1 + 1e-9)
        if graph.startswith(('swin', 'logwin')) and world == 8:
            assert max(plan.cost) < 0.7 * max(contiguous)         # 600 hash-ordered pairs: a contiguous slice touches ~3x the images it needs
    # pair-by-pair engines (no encode-once): equal pair counts in list order
    plan = shard_plan(pairs, 4, encode_once=False)
    assert plan.name == 'list order' and max(plan.counts) == -(-P // 4) and plan.order == list(range(P))


def test_shard_plan_weights_mixed_image_sizes_by_area():
    from dust3r_amd.parallel import shard_plan
    pairs = _mixed_pairs()
    plan = shard_plan(pairs, 3, encode_once=False)
    assert sorted(k for r in range(3) for k in plan.shard(r)) == list(range(len(pairs))) and max(plan.cost) / min(plan.cost) < 1.35


def test_image_ranges_of_the_rank_shared_alignment_loop():
    """cloud_opt.base_opt.image_ranges (host logic of compute_global_alignment(group=...)): contiguous, disjoint, covering ranges; balanced by area x (incident edge sides + 1.5);
    more ranks than images leaves some ranks empty; one rank owns everything."""
    from dust3r_amd.cloud_opt.base_opt import image_ranges
    from dust3r_amd.synthetic import scene_edges
    for n, graph, sym, world in ((20, 'complete', False, 8), (100, 'swin-3', True, 8), (3, 'complete', True, 8), (20, 'complete', False, 1), (7, 'complete', True, 2), (11, 'swin-2', True, 3)):
        edges = scene_edges(n, graph, sym)
        shapes = [(384, 512)] * n
        r = image_ranges(edges, shapes, world)
        assert len(r) == world and r[0][0] == 0 and sum(c for _, c in r) == n and all(c >= 0 for _, c in r)
        assert all(r[k][0] + r[k][1] == r[k + 1][0] for k in range(world - 1))
        if n >= 2 * world:                                      # uniform graphs: no rank more than one image above another
            counts = [c for _, c in r]
            assert max(counts) - min(counts) <= 1, (n, graph, world, counts)
    # mixed sizes and degrees: a star graph (image 0 carries every edge) with one more large image -- the heavy image gets a rank of its own
    edges = [(0, k) for k in range(1, 9)] + [(k, 0) for k in range(1, 9)]
    shapes = [(384, 512)] + [(96, 128)] * 7 + [(384, 512)]
    r = image_ranges(edges, shapes, 3)
    assert sum(c for _, c in r) == 9 and [c for f, c in r if f <= 0 < f + c] == [1]       # image 0 alone on its rank (the bound of the job: it cannot be split)


def test_upload_helpers_fall_back_off_gpu():
    """utils/device.py upload_rows / upload_stack with a non-CUDA target (and small or non-tensor inputs): plain .to() semantics."""
    from dust3r_amd.utils.device import upload_rows
    t = torch.arange(24, dtype=torch.float32).reshape(4, 6)
    assert torch.equal(upload_rows(t, 'cpu'), t) and upload_rows('x', 'cpu') == 'x'
    big = torch.rand(40, 256, 256)                            # 10 MB: above the piece size, still the plain route on a CPU target
    assert torch.equal(upload_rows(big, torch.device('cpu')), big)
