"""Inference driver -- mirror of the reference `dust3r/inference.py:26-78` (`inference`,
`loss_of_one_batch` with criterion=None, `check_if_same_size`, `make_batch_symmetric`).

Same signature and same returned structure: dict(view1, view2, pred1, pred2, loss=None) with every
tensor on the CPU, concatenated over pairs (lists when image sizes are mixed). The model call goes
to the HIP engine; pairs can additionally be sharded over ranks with `dust3r_amd.parallel`.
"""
import torch
import tqdm

from .utils.device import collate_with_cat, to_cpu


def _interleave_imgs(img1, img2):
    res = {}
    for key, value1 in img1.items():
        value2 = img2[key]
        if isinstance(value1, torch.Tensor):
            res[key] = torch.stack((value1, value2), dim=1).flatten(0, 1)
        else:
            res[key] = [x for pair in zip(value1, value2) for x in pair]
    return res


def make_batch_symmetric(batch):
    view1, view2 = batch
    return _interleave_imgs(view1, view2), _interleave_imgs(view2, view1)


def loss_of_one_batch(batch, model, criterion, device, symmetrize_batch=False, use_amp=False, ret=None):
    assert criterion is None, 'training losses are outside the scope of dust3r_amd (inference + alignment engine)'
    view1, view2 = batch
    ignore_keys = set(['depthmap', 'dataset', 'label', 'instance', 'idx', 'true_shape', 'rng'])
    for view in batch:
        for name in view.keys():
            if name in ignore_keys:
                continue
            view[name] = view[name].to(device, non_blocking=True)
    if symmetrize_batch:
        view1, view2 = make_batch_symmetric(batch)
    pred1, pred2 = model(view1, view2)      # use_amp is moot: the engine's precision is a model property
    result = dict(view1=view1, view2=view2, pred1=pred1, pred2=pred2, loss=None)
    return result[ret] if ret else result


def check_if_same_size(pairs):
    shapes1 = [img1['img'].shape[-2:] for img1, img2 in pairs]
    shapes2 = [img2['img'].shape[-2:] for img1, img2 in pairs]
    return all(shapes1[0] == s for s in shapes1) and all(shapes2[0] == s for s in shapes2)


def _encode_once_ok(pairs, model):
    """The encode-once path needs an engine with encode/decode entry points, one image size, and an `idx` per view that names
    the image (what load_images / make_pairs produce): then each distinct image is encoded once instead of once per pair."""
    if not hasattr(model, 'encode_images') or getattr(model, '_engine', None) is None:
        return False
    try:
        ids = {}
        for v1, v2 in pairs:
            for v in (v1, v2):
                if v['img'].shape[0] != 1:
                    return False
                key = int(v['idx'])
                if key in ids and ids[key] is not v['img'] and not torch.equal(ids[key], v['img']):
                    return False            # same idx, different pixels: not an image id
                ids[key] = v['img']
        return len(ids) < 2 * len(pairs)    # nothing shared: the plain path does the same work
    except (KeyError, TypeError, ValueError):
        return False


def _alloc_outputs(P, H, W, out_device):
    kw = dict(dtype=torch.float32, device=out_device)
    return (dict(pts3d=torch.empty((P, H, W, 3), **kw), conf=torch.empty((P, H, W), **kw)),
            dict(pts3d_in_other_view=torch.empty((P, H, W, 3), **kw), conf=torch.empty((P, H, W), **kw)))


class _PredictionSink:
    """Where the per-batch predictions go. Device outputs: a plain copy. Host outputs (the reference's format, inference.py:68): a ring
    of two PINNED staging sets; batch k's D2H copies run on a side stream while batch k + 1 computes, and the host moves batch k - 1
    from the ring into the (pageable) result tensors meanwhile -- the GPU never waits for PCIe or for the host memcpy, and the big
    result tensors need no pinned allocation."""

    KEYS = (('pred1', 'pts3d'), ('pred1', 'conf'), ('pred2', 'pts3d_in_other_view'), ('pred2', 'conf'))

    def __init__(self, n_pairs, H, W, batch_size, out_device, compute_device):
        self.pred1, self.pred2 = _alloc_outputs(n_pairs, H, W, out_device)
        self.out = dict(pred1=self.pred1, pred2=self.pred2)
        self.host = torch.device(out_device).type == 'cpu' and torch.device(compute_device).type == 'cuda'
        self.pending = []
        if self.host:
            self.stream = torch.cuda.Stream(device=compute_device)
            shapes = {('pred1', 'pts3d'): (batch_size, H, W, 3), ('pred1', 'conf'): (batch_size, H, W),
                      ('pred2', 'pts3d_in_other_view'): (batch_size, H, W, 3), ('pred2', 'conf'): (batch_size, H, W)}
            self.ring = [{k: torch.empty(shapes[k], dtype=torch.float32, pin_memory=True) for k in self.KEYS} for _ in range(2)]
            self.slot = 0

    def put(self, i, j, p1, p2):
        src = {('pred1', 'pts3d'): p1['pts3d'], ('pred1', 'conf'): p1['conf'], ('pred2', 'pts3d_in_other_view'): p2['pts3d_in_other_view'],
               ('pred2', 'conf'): p2['conf']}
        if not self.host:
            for (a, b), t in src.items():
                self.out[a][b][i:j].copy_(t, non_blocking=True)
            return
        self._drain(keep=1)                                  # the slot about to be reused has been emptied by the host
        ring = self.ring[self.slot]
        self.stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.stream):
            for k, t in src.items():
                ring[k][:j - i].copy_(t, non_blocking=True)
                t.record_stream(self.stream)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        self.pending.append((ev, self.slot, i, j))
        self.slot ^= 1

    def _drain(self, keep=0):
        while len(self.pending) > keep:
            ev, slot, i, j = self.pending.pop(0)
            ev.synchronize()
            for (a, b) in self.KEYS:
                self.out[a][b][i:j].copy_(self.ring[slot][(a, b)][:j - i])

    def finish(self):
        if self.host:
            self._drain(keep=0)
        elif torch.cuda.is_available():
            torch.cuda.synchronize()
        return self.pred1, self.pred2


def _collate_views(pairs):
    """view1 / view2 dicts of the whole pair list in the reference's collated format (inference.py:68-72: tensors concatenated on the
    host, lists chained). When the pair list shares images (every `img` tensor object appears in several pairs) the big `img` tensors are
    built by ONE multi-threaded gather from the stack of distinct images instead of a serial concatenation of 2 x len(pairs) pieces."""
    uniq, index = {}, ([], [])
    for side in (0, 1):
        for p in pairs:
            t = p[side]['img']
            index[side].append(uniq.setdefault(id(t), (len(uniq), t))[0])
    if len(uniq) < len(pairs) and all(t.shape[0] == 1 and not t.is_cuda for _, t in uniq.values()):
        stack = torch.cat([t for _, t in sorted(uniq.values(), key=lambda x: x[0])], dim=0)
        light = [tuple({k: v for k, v in view.items() if k != 'img'} for view in p) for p in pairs]
        view1, view2 = collate_with_cat(light)
        view1['img'] = stack.index_select(0, torch.tensor(index[0]))
        view2['img'] = stack.index_select(0, torch.tensor(index[1]))
        return view1, view2
    return collate_with_cat(list(pairs))


class _Background:
    """Run a host-only function on a thread while the GPU loop runs (torch releases the GIL inside its kernels)."""

    def __init__(self, fn, *args):
        import threading
        self.result, self.error = None, None

        def run():
            try:
                self.result = fn(*args)
            except BaseException as e:     # re-raised in join()
                self.error = e
        self.thread = threading.Thread(target=run, daemon=True)
        self.thread.start()

    def join(self):
        self.thread.join()
        if self.error is not None:
            raise self.error
        return self.result


@torch.no_grad()
def inference_encode_once(pairs, model, device, batch_size=8, verbose=True, output_device='cpu'):
    """Same return value as `inference` (bit-identical: every engine kernel is batch-position independent), but every distinct
    image goes through the ViT-L encoder ONCE: n encoder passes instead of 2 x len(pairs) -- 20 instead of 380 for the demo's
    complete symmetrised graph over 20 views, i.e. ~53 % fewer FLOPs end to end. Predictions stream to the host
    behind the compute (see _PredictionSink); the view dicts are collated on a host thread meanwhile."""
    views = _Background(_collate_views, pairs)
    imgs, order = {}, []
    for v1, v2 in pairs:
        for v in (v1, v2):
            k = int(v['idx'])
            if k not in imgs:
                imgs[k] = v['img']
                order.append(k)
    pos = {k: i for i, k in enumerate(order)}
    H, W = pairs[0][0]['img'].shape[-2:]
    feats = []
    enc_bs = max(2, 2 * batch_size)
    for i in tqdm.trange(0, len(order), enc_bs, disable=not verbose, desc='encode'):
        feats.append(model.encode_images(torch.cat([imgs[k] for k in order[i:i + enc_bs]], dim=0).to(device, non_blocking=True)))
    feats = torch.cat(feats, dim=0)
    sink = _PredictionSink(len(pairs), H, W, batch_size, output_device, feats.device)
    i1 = torch.tensor([pos[int(a['idx'])] for a, _ in pairs], device=feats.device)
    i2 = torch.tensor([pos[int(b['idx'])] for _, b in pairs], device=feats.device)
    for i in tqdm.trange(0, len(pairs), batch_size, disable=not verbose, desc='decode'):
        j = min(i + batch_size, len(pairs))
        p1, p2 = model.decode_pairs(feats.index_select(0, torch.cat((i1[i:j], i2[i:j]))), H, W)
        sink.put(i, j, p1, p2)
    pred1, pred2 = sink.finish()
    view1, view2 = views.join()
    return dict(view1=view1, view2=view2, pred1=pred1, pred2=pred2, loss=None)


@torch.no_grad()
def inference(pairs, model, device, batch_size=8, verbose=True, encode_once=None, output_device='cpu'):
    """Mirror of dust3r/inference.py:55-72. Extras (defaults keep the reference's behaviour): `encode_once` (None = automatic:
    encode every distinct image once when the pair list shares images) and `output_device` ('cpu' like the reference's
    to_cpu; a CUDA device keeps the predictions in HBM for `global_aligner(output, device)`, which would upload them again)."""
    if verbose:
        print(f'>> Inference with model on {len(pairs)} image pairs')
    multiple_shapes = not check_if_same_size(pairs)
    if encode_once is None:
        encode_once = True
    if encode_once and not multiple_shapes and _encode_once_ok(pairs, model):
        return inference_encode_once(pairs, model, device, batch_size=batch_size, verbose=verbose, output_device=output_device)
    if multiple_shapes:
        result = []
        for i in tqdm.trange(0, len(pairs), 1, disable=not verbose):
            res = loss_of_one_batch(collate_with_cat(pairs[i:i + 1]), model, None, device)
            result.append(to_cpu(res) if str(output_device) == 'cpu' else res)
        return collate_with_cat(result, lists=True)
    views = _Background(_collate_views, pairs)
    H, W = pairs[0][0]['img'].shape[-2:]
    sink = _PredictionSink(len(pairs), H, W, batch_size, output_device, device)
    for i in tqdm.trange(0, len(pairs), batch_size, disable=not verbose):
        j = min(i + batch_size, len(pairs))
        res = loss_of_one_batch(collate_with_cat(pairs[i:j]), model, None, device)
        sink.put(i, j, res['pred1'], res['pred2'])
    pred1, pred2 = sink.finish()
    view1, view2 = views.join()
    return dict(view1=view1, view2=view2, pred1=pred1, pred2=pred2, loss=None)
