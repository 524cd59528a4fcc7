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


def _pairs(n_views, H, W):
    sys.path.insert(0, ROOT)
    from dust3r_amd.image_pairs import make_pairs
    from dust3r_amd.synthetic import synthetic_image_list
    return make_pairs(synthetic_image_list(n_views, H, W, seed=3), 'complete', None, symmetrize=False)


def _worker(rank, world, port, n_views, outdir, H=16, W=32):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from dust3r_amd.parallel import inference_sharded
        out = inference_sharded(_pairs(n_views, H, W), StandInModel(), 'cpu', batch_size=2)
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
