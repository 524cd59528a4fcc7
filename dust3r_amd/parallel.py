"""Multi-GPU inference: one process per GPU, pairs sharded across ranks, ONE all-gather.

The reference's `inference()` (dust3r/inference.py:55-72) is a single-device loop whose
iterations share no state, so image pairs shard embarrassingly: rank r runs the
engine on a contiguous slice of the `make_pairs` list with replicated weights, and the pairwise
predictions (pred1.pts3d, pred1.conf, pred2.pts3d_in_other_view, pred2.conf: 8 fp32 per pixel)
are collected with a single `all_gather_into_tensor` -- RCCL over xGMI on the GPU box
(backend "nccl"), gloo in the CPU tests. Every rank ends up with the dict that single-device
`inference()` returns, ready for `global_aligner` (which BASELINE.json runs on one GPU).

Nothing here depends on the device type: the gloo tests drive the same code with a stand-in model.
"""
import torch
import torch.distributed as dist

from .utils.device import collate_with_cat


def shard_bounds(n_items, rank, world_size):
    """Contiguous shard [lo, hi) of rank `rank`; every shard has ceil(n/world) slots (the tail is padding)."""
    per = (n_items + world_size - 1) // world_size
    lo = min(rank * per, n_items)
    return lo, min(lo + per, n_items), per


def pack_predictions(pred1, pred2):
    """(P,H,W,3),(P,H,W),(P,H,W,3),(P,H,W) -> one (P,H,W,8) fp32 tensor: a single collective payload."""
    return torch.cat((pred1['pts3d'], pred1['conf'][..., None], pred2['pts3d_in_other_view'], pred2['conf'][..., None]), dim=-1).contiguous()


def unpack_predictions(packed):
    pred1 = dict(pts3d=packed[..., 0:3].contiguous(), conf=packed[..., 3].contiguous())
    pred2 = dict(pts3d_in_other_view=packed[..., 4:7].contiguous(), conf=packed[..., 7].contiguous())
    return pred1, pred2


def all_gather_packed(local, group=None, async_op=False, out=None):
    """All-gather equal-sized per-rank payloads (per, ...) -> (world*per, ...). The one collective of the path."""
    world = dist.get_world_size(group)
    if out is None:
        out = local.new_empty((world * local.shape[0],) + tuple(local.shape[1:]))
    work = dist.all_gather_into_tensor(out, local, group=group, async_op=async_op)
    return (out, work) if async_op else out


@torch.no_grad()
def inference_sharded(pairs, model, device, batch_size=8, verbose=False, group=None, gather_device=None):
    """Drop-in for `inference(pairs, model, device, batch_size)` when torch.distributed is initialised:
    identical return value on every rank. Requires all pairs to share one image size (the sharded
    path is the throughput path; mixed sizes go through `inference`)."""
    from .inference import _engine_step, check_if_same_size, loss_of_one_batch
    assert check_if_same_size(pairs), 'inference_sharded needs pairs of one image size'
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    lo, hi, per = shard_bounds(len(pairs), rank, world)
    batch_size = _engine_step(model, batch_size)     # at least the engine's preferred pairs per call (bit-identical results)
    gather_device = torch.device(gather_device if gather_device is not None else device)
    H, W = pairs[0][0]['img'].shape[-2:]
    local = torch.zeros((per, H, W, 8), dtype=torch.float32, device=gather_device)
    for i in range(lo, hi, batch_size):
        batch = collate_with_cat(pairs[i:min(i + batch_size, hi)])
        n = batch[0]['img'].shape[0]
        if hasattr(model, 'forward_packed') and local.is_cuda:
            # the engine's heads write the interleaved payload in place: no pack pass
            model.forward_packed(dict(img=batch[0]['img'].to(device)), dict(img=batch[1]['img'].to(device)), out=local[i - lo:i - lo + n])
        else:
            res = loss_of_one_batch(batch, model, None, device)
            local[i - lo:i - lo + n] = pack_predictions(res['pred1'], res['pred2']).to(gather_device)
    gathered = all_gather_packed(local, group)
    # drop the padding slots of the short last shards
    keep = torch.cat([torch.arange(r * per, r * per + (shard_bounds(len(pairs), r, world)[1] - shard_bounds(len(pairs), r, world)[0]))
                      for r in range(world)]).to(gathered.device)
    pred1, pred2 = unpack_predictions(gathered.index_select(0, keep).cpu())
    # view metadata is rebuilt deterministically on every rank (host side), as SURVEY.md 8(e) prescribes
    view1 = collate_with_cat([(p[0], p[1]) for p in pairs])[0]
    view2 = collate_with_cat([(p[0], p[1]) for p in pairs])[1]
    view1 = {k: (v.cpu() if isinstance(v, torch.Tensor) else v) for k, v in view1.items()}
    view2 = {k: (v.cpu() if isinstance(v, torch.Tensor) else v) for k, v in view2.items()}
    return dict(view1=view1, view2=view2, pred1=pred1, pred2=pred2, loss=None)
