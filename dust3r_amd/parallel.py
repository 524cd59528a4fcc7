"""Multi-GPU inference: one process per GPU, pairs sharded across ranks, ONE all-gather.

The reference's `inference()` (dust3r/inference.py:55-72) is a single-device loop whose
iterations share no state, so image pairs shard embarrassingly: rank r runs the
engine on a contiguous slice of the `make_pairs` list with replicated weights, and the pairwise
predictions (pred1.pts3d, pred1.conf, pred2.pts3d_in_other_view, pred2.conf: 8 fp32 per pixel)
are collected with a single `all_gather_into_tensor` -- RCCL over xGMI on the GPU box
(backend "nccl"), gloo in the CPU tests. Every rank ends up with the dict that single-device
`inference()` returns, ready for `global_aligner` (which BASELINE.json runs on one GPU).

Every rank encodes only the distinct images its own shard touches (encode-once, like single-device `inference()`), and pair lists of
several image sizes shard too (flat padded payload). Nothing here depends on the device type: the gloo tests drive the same code with
a stand-in model.

Which pairs a rank runs is decided by `shard_plan` (round 5): the pair list is put in an order that keeps the pairs of an image together
(the reference's windowed graphs come out of a `set`, image_pairs.py:17-33, i.e. in hash order: a contiguous slice of THAT list touches
almost every image), then cut where max over ranks of (encoder passes + decoder/head passes) is smallest. The gathered rows are put back
in the caller's order by the same index_select that used to drop the padding rows, so the result is unchanged: still ONE collective.
"""
import torch
import torch.distributed as dist

from .utils.device import collate_with_cat, upload_stack


def shard_bounds(n_items, rank, world_size):
    """Contiguous shard [lo, hi) of rank `rank`; every shard has ceil(n/world) slots (the tail is padding)."""
    per = (n_items + world_size - 1) // world_size
    lo = min(rank * per, n_items)
    return lo, min(lo + per, n_items), per


# SURVEY.md 8(d), GFLOP at 512x384: one encoder pass per distinct image of a shard, both decoders + both DPT heads per pair. Only the RATIO
# matters to the plan (it scales with the pixel count on both sides).
ENC_COST_PER_IMAGE = 523.0
DEC_COST_PER_PAIR = 2 * 218.6 + 2 * 186.7


def _morton(a, b):
    z = 0
    for k in range(20):
        z |= ((a >> k) & 1) << (2 * k + 1) | ((b >> k) & 1) << (2 * k)
    return z


class ShardPlan:
    """order: pair indices in the order the ranks' shards concatenate; bounds[r] = (lo, hi) into `order`; per = rows of the all-gather
    payload per rank (the longest shard; shorter ones are zero padded); source[k] = row of the gathered (world * per) payload that holds
    pair k of the caller's list; cost[r] = modelled GFLOP of rank r; images[r] = distinct images rank r encodes."""

    def __init__(self, order, bounds, cost, images, name):
        self.order, self.bounds, self.cost, self.images, self.name = order, bounds, cost, images, name
        self.world = len(bounds)
        self.per = max([hi - lo for lo, hi in bounds] + [0])
        self.counts = [hi - lo for lo, hi in bounds]
        src = [0] * len(order)
        for r, (lo, hi) in enumerate(bounds):
            for t in range(lo, hi):
                src[order[t]] = r * self.per + (t - lo)
        self.source = torch.tensor(src, dtype=torch.long)

    def shard(self, rank):
        lo, hi = self.bounds[rank]
        return self.order[lo:hi]

    def summary(self):
        return dict(order=self.name, pairs_per_rank=self.counts, distinct_images_per_rank=self.images, modelled_gflop_per_rank=[round(c, 1) for c in self.cost],
                    imbalance_max_over_mean=(max(self.cost) / (sum(self.cost) / len(self.cost)) if sum(self.cost) > 0 else 1.0))


def _balanced_cuts(order, edges, area, world, enc_cost):
    """Contiguous cuts of `order` into <= world shards minimising max over shards of enc_cost * (area-weighted distinct images) + pair
    cost: bisection on the bound, greedy feasibility (a shard's cost only grows when it is extended, so the greedy sweep is exact)."""
    P = len(order)
    pair_cost = [DEC_COST_PER_PAIR * 0.5 * (area[edges[k][0]] + area[edges[k][1]]) for k in order]

    def sweep(bound):
        cuts, lo = [], 0
        while lo < P:
            if len(cuts) == world:
                return None
            seen, c, hi = set(), 0.0, lo
            while hi < P:
                add = pair_cost[hi] + sum(enc_cost * area[v] for v in set(edges[order[hi]]) if v not in seen)
                if c + add > bound:
                    break
                seen.update(edges[order[hi]])
                c += add
                hi += 1
            if hi == lo:
                return None
            cuts.append((lo, hi, c, len(seen)))
            lo = hi
        return cuts

    total = sum(pair_cost) + enc_cost * sum(area[v] for v in {v for e in edges for v in e})
    lo_b, hi_b = 0.0, total * (1 + 1e-9) + 1e-9
    for _ in range(48):
        mid = 0.5 * (lo_b + hi_b)
        if sweep(mid) is None:
            lo_b = mid
        else:
            hi_b = mid
    cuts = sweep(hi_b)
    cuts += [(P, P, 0.0, 0)] * (world - len(cuts))
    return cuts


def shard_plan(pairs, world, encode_once=True):
    """The assignment of pairs to ranks (same on every rank: a pure function of the pair list's image indices and sizes).
    Candidates: the caller's order, pairs sorted by (smaller image index, larger image index) -- (i, j) next to (j, i), an image's
    window together -- and the Z-order curve over (larger, smaller) index (runs of it are square blocks of a dense graph's adjacency
    matrix: ~2 sqrt(pairs) images per shard instead of a whole row). Each is cut by `_balanced_cuts`; the smallest maximum wins."""
    edges = [(int(a['idx']), int(b['idx'])) for a, b in pairs]
    P = len(edges)
    area = {}
    for (a, b), (i, j) in zip(pairs, edges):
        for v, k in ((a, i), (b, j)):
            if k not in area:
                img = v.get('img') if isinstance(v, dict) else None
                area[k] = float(img.shape[-2] * img.shape[-1]) if img is not None else 1.0
    if area:
        ref_area = max(area.values())
        area = {k: a / ref_area for k, a in area.items()}
    enc_cost = ENC_COST_PER_IMAGE if encode_once else 0.0
    cands = [('list order', list(range(P)))]
    if encode_once and world > 1:
        cands.append(('sorted by (min, max) image index', sorted(range(P), key=lambda k: (min(edges[k]), max(edges[k]), k))))
        cands.append(('Z-order over (max, min) image index', sorted(range(P), key=lambda k: (_morton(max(edges[k]), min(edges[k])), k))))
    best = None
    for name, order in cands:
        cuts = _balanced_cuts(order, edges, area, world, enc_cost)
        worst = max(c[2] for c in cuts) if cuts else 0.0
        if best is None or worst < best[0] * (1 - 1e-12):
            best = (worst, name, order, cuts)
    _, name, order, cuts = best
    return ShardPlan(order, [(lo, hi) for lo, hi, _, _ in cuts], [c for _, _, c, _ in cuts], [n for _, _, _, n in cuts], name)


def pack_predictions(pred1, pred2):
    """(P,H,W,3),(P,H,W),(P,H,W,3),(P,H,W) -> one (P,H,W,8) fp32 tensor: a single collective payload."""
    return torch.cat((pred1['pts3d'], pred1['conf'][..., None], pred2['pts3d_in_other_view'], pred2['conf'][..., None]), dim=-1).contiguous()


def unpack_predictions(packed):
    pred1 = dict(pts3d=packed[..., 0:3].contiguous(), conf=packed[..., 3].contiguous())
    pred2 = dict(pts3d_in_other_view=packed[..., 4:7].contiguous(), conf=packed[..., 7].contiguous())
    return pred1, pred2


def _unpack_to_host(gathered, keep, chunk=64):
    """The gathered packed payload (rows, H, W, 8) -> the caller's pair order, as the four result tensors of `inference()` ON THE HOST. From a GPU: the rows
    are selected and the four tensors sliced on the device, chunk by chunk, and each piece is copied into host result memory on huge pages
    (utils/device.py:host_tensor) -- `.cpu()` of the whole payload followed by four strided host copies moved the 3.8 GB of 600 pairs three times."""
    if not gathered.is_cuda:
        return unpack_predictions(gathered.index_select(0, keep.to(gathered.device)))
    from .inference import _alloc_outputs
    P, (H, W) = len(keep), gathered.shape[1:3]
    pred1, pred2 = _alloc_outputs(P, H, W, 'cpu')
    keep_d = keep.to(gathered.device)
    for i in range(0, P, chunk):
        rows = gathered.index_select(0, keep_d[i:i + chunk])
        pred1['pts3d'][i:i + chunk].copy_(rows[..., 0:3].contiguous())
        pred1['conf'][i:i + chunk].copy_(rows[..., 3].contiguous())
        pred2['pts3d_in_other_view'][i:i + chunk].copy_(rows[..., 4:7].contiguous())
        pred2['conf'][i:i + chunk].copy_(rows[..., 7].contiguous())
    return pred1, pred2


def all_gather_packed(local, group=None, async_op=False, out=None):
    """All-gather equal-sized per-rank payloads (per, ...) -> (world*per, ...). The one collective of the path."""
    world = dist.get_world_size(group)
    if out is None:
        out = local.new_empty((world * local.shape[0],) + tuple(local.shape[1:]))
    work = dist.all_gather_into_tensor(out, local, group=group, async_op=async_op)
    return (out, work) if async_op else out


def _pair_shapes(pairs):
    return [(tuple(int(x) for x in a['img'].shape[-2:]), tuple(int(x) for x in b['img'].shape[-2:])) for a, b in pairs]


def _flat_len(hw1, hw2):
    """floats of one pair's predictions in the flat payload [pts1 (A1 x 3) | conf1 (A1) | pts2 (A2 x 3) | conf2 (A2)]"""
    return 4 * (hw1[0] * hw1[1] + hw2[0] * hw2[1])


def _flatten_rows(pred1, pred2, n):
    """(n, H1, W1, 3), (n, H1, W1), (n, H2, W2, 3), (n, H2, W2) -> (n, 4 (A1 + A2)) in the flat payload order."""
    return torch.cat((pred1['pts3d'].reshape(n, -1), pred1['conf'].reshape(n, -1), pred2['pts3d_in_other_view'].reshape(n, -1),
                      pred2['conf'].reshape(n, -1)), dim=1)


def _unflatten_row(row, hw1, hw2):
    a1, a2 = hw1[0] * hw1[1], hw2[0] * hw2[1]
    o = [0, 3 * a1, 4 * a1, 4 * a1 + 3 * a2, 4 * (a1 + a2)]
    return (dict(pts3d=row[o[0]:o[1]].reshape(1, *hw1, 3).clone(), conf=row[o[1]:o[2]].reshape(1, *hw1).clone()),
            dict(pts3d_in_other_view=row[o[2]:o[3]].reshape(1, *hw2, 3).clone(), conf=row[o[3]:o[4]].reshape(1, *hw2).clone()))


def _local_same_size(shard, per, model, device, batch_size, gather_device, encode_once, H, W):
    """This rank's shard (a list of pairs) of a one-size pair list -> (per, H, W, 8) packed payload (zero in the padding slots)."""
    from .inference import _encode_once_ok, loss_of_one_batch
    local = torch.zeros((per, H, W, 8), dtype=torch.float32, device=gather_device)
    if not shard:
        return local
    if encode_once and _encode_once_ok(shard, model):
        # every DISTINCT image of the shard goes through the encoder once (contiguous slices of make_pairs' list share images: the
        # complete graph over 20 views cut in 8 touches 4-9 images per rank instead of 2 x 24 pair slots), then the shard's pairs are decoded
        imgs, order = {}, []
        for v1, v2 in shard:
            for v in (v1, v2):
                k = int(v['idx'])
                if k not in imgs:
                    imgs[k] = v['img']
                    order.append(k)
        pos = {k: i for i, k in enumerate(order)}
        enc_bs = max(2, 2 * batch_size)
        feats = torch.cat([model.encode_images(upload_stack([imgs[k] for k in order[i:i + enc_bs]], device)) for i in range(0, len(order), enc_bs)], dim=0)
        i1 = torch.tensor([pos[int(a['idx'])] for a, _ in shard], device=feats.device)
        i2 = torch.tensor([pos[int(b['idx'])] for _, b in shard], device=feats.device)
        for i in range(0, len(shard), batch_size):
            j = min(i + batch_size, len(shard))
            f = feats.index_select(0, torch.cat((i1[i:j], i2[i:j])))
            if local.is_cuda and local.device == feats.device:
                model.decode_pairs(f, H, W, packed_out=local[i:j])           # the heads write the payload in place
            else:
                p1, p2 = model.decode_pairs(f, H, W)
                local[i:j] = pack_predictions(p1, p2).to(gather_device)
        return local
    for i in range(0, len(shard), batch_size):
        batch = collate_with_cat(shard[i:i + batch_size])
        n = batch[0]['img'].shape[0]
        if hasattr(model, 'forward_packed') and local.is_cuda:
            # the engine's heads write the interleaved payload in place: no pack pass
            model.forward_packed(dict(img=batch[0]['img'].to(device)), dict(img=batch[1]['img'].to(device)), out=local[i:i + n])
        else:
            res = loss_of_one_batch(batch, model, None, device)
            local[i:i + n] = pack_predictions(res['pred1'], res['pred2']).to(gather_device)
    return local


def _local_mixed(pairs, shapes, mine, per, slot, model, device, batch_size, gather_device):
    """This rank's shard of a pair list with SEVERAL image sizes -> (per, slot) flat payload: the shard's pairs are grouped by their two
    image sizes and every group is batched (dust3r_amd.inference does the same on one device); slot = the longest pair of the WHOLE list."""
    from .inference import loss_of_one_batch
    local = torch.zeros((per, slot), dtype=torch.float32, device=gather_device)
    groups, row_of = {}, {k: t for t, k in enumerate(mine)}
    for k in mine:
        groups.setdefault(shapes[k], []).append(k)
    for (hw1, hw2), members in groups.items():
        L = _flat_len(hw1, hw2)
        for i in range(0, len(members), batch_size):
            chunk = members[i:i + batch_size]
            res = loss_of_one_batch(collate_with_cat([pairs[k] for k in chunk]), model, None, device)
            rows = _flatten_rows(res['pred1'], res['pred2'], len(chunk)).to(gather_device)
            local[torch.tensor([row_of[k] for k in chunk], device=gather_device), :L] = rows
    return local


@torch.no_grad()
def inference_sharded(pairs, model, device, batch_size=8, verbose=False, group=None, gather_device=None, encode_once=None, engine_batch=None):
    """Drop-in for `inference(pairs, model, device, batch_size)` when torch.distributed is initialised: identical return value on
    every rank (bit-identical to the single-process call: every engine kernel is batch-position independent). Rank r runs a contiguous
    slice of the pair list; the predictions travel in ONE all_gather_into_tensor.
      * one image size: the payload is the engine's packed (pairs, H, W, 8) tensor, written in place by the head epilogues; with
        `encode_once` (None = automatic, like inference()) a rank encodes every distinct image of ITS shard once and decodes its pairs;
      * several image sizes (a portrait picture among landscape ones, dust3r/inference.py:60-68): pairs are grouped by shape inside
        the shard and the payload is one flat fp32 row per pair, padded to the longest pair of the list -- still one collective; the
        result has the reference's list-per-pair structure, exactly as `inference()` returns it for such a list."""
    from .inference import _engine_step, check_if_same_size
    from .utils.device import to_cpu
    from .inference import _encode_once_ok
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    batch_size = _engine_step(model, batch_size, engine_batch)     # at least the engine's preferred pairs per call (bit-identical results)
    gather_device = torch.device(gather_device if gather_device is not None else device)
    same = check_if_same_size(pairs)
    enc1 = (encode_once is None or bool(encode_once)) and same and len(pairs) > 0 and _encode_once_ok(pairs, model)
    plan = shard_plan(pairs, world, encode_once=enc1)
    mine, per = plan.shard(rank), plan.per
    keep = plan.source                                             # caller's order <- gathered rows (also drops the padding rows of short shards)
    if same:
        H, W = pairs[0][0]['img'].shape[-2:]
        from .inference import _Background, _collate_views, _shared_images
        # the collated view images of the result (rebuilt deterministically on every rank) leave for the host on a thread of their own
        # while the shard runs: gathered on the GPU from the distinct images when the pair list shares them, like inference()
        views = None
        shared = _shared_images(pairs) if torch.device(device).type == 'cuda' else None
        if shared is not None:
            stack = upload_stack(shared[0], device)
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream(stack.device))
            views = _Background(_collate_views, pairs, shared, stack, ready)
        local = _local_same_size([pairs[k] for k in mine], per, model, device, batch_size, gather_device, enc1, H, W)
        gathered = all_gather_packed(local, group)
        pred1, pred2 = _unpack_to_host(gathered, keep)
        view1, view2 = views.join() if views is not None else collate_with_cat([(p[0], p[1]) for p in pairs])
        view1 = {k: (v.cpu() if isinstance(v, torch.Tensor) else v) for k, v in view1.items()}
        view2 = {k: (v.cpu() if isinstance(v, torch.Tensor) else v) for k, v in view2.items()}
        return dict(view1=view1, view2=view2, pred1=pred1, pred2=pred2, loss=None)
    shapes = _pair_shapes(pairs)
    slot = max(_flat_len(*s) for s in shapes)
    local = _local_mixed(pairs, shapes, mine, per, slot, model, device, batch_size, gather_device)
    gathered = all_gather_packed(local, group).index_select(0, keep.to(gather_device)).cpu()
    result = []
    for k, (a, b) in enumerate(pairs):
        v1, v2 = to_cpu(collate_with_cat([(a, b)]))
        p1, p2 = _unflatten_row(gathered[k], *shapes[k])
        result.append(dict(view1=v1, view2=v2, pred1=p1, pred2=p2, loss=None))
    return collate_with_cat(result, lists=True)
