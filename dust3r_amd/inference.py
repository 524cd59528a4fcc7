"""Inference driver -- mirror of the reference `dust3r/inference.py:26-78` (`inference`,
`loss_of_one_batch` with criterion=None, `check_if_same_size`, `make_batch_symmetric`).

Same signature and same returned structure: dict(view1, view2, pred1, pred2, loss=None) with every
tensor on the CPU, concatenated over pairs (lists when image sizes are mixed). The model call goes
to the HIP engine; pairs can additionally be sharded over ranks with `dust3r_amd.parallel`.
"""
import torch
import tqdm

from .utils.device import collate_with_cat, host_tensor, to_cpu, upload_stack


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


def _engine_step(model, batch_size, engine_batch=None):
    """Pairs per model call. The reference runs `batch_size` pairs per call; an engine that declares `engine_batch` (default 32, or
    DUST3R_AMD_ENGINE_BATCH) gets at least that many -- the demo names 1, which is launch-bound on a GPU -- because its results do not
    depend on how the pair list is cut into batches (bit-identical, tests/test_forward_gpu.py). A caller who needs SMALLER calls (larger
    images, less free HBM: the workspace grows with the batch) passes `engine_batch=n` to inference(): exactly n pairs per call."""
    if engine_batch is not None:
        return max(int(engine_batch), 1)
    return max(int(batch_size), int(getattr(model, 'engine_batch', 0) or 0), 1)


def _rows_of(res, r):
    """Row r of a collated result (nested dicts of tensors / lists), as the one-pair batch the reference's per-pair loop produces."""
    if isinstance(res, dict):
        return {k: _rows_of(v, r) for k, v in res.items()}
    if isinstance(res, torch.Tensor):
        return res[r:r + 1]
    if isinstance(res, (list, tuple)):
        return type(res)([res[r]])
    return res


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
    """The four result tensors. On the host: zero-filled and touched, on huge pages when they are large (utils/device.py:host_tensor)."""
    if torch.device(out_device).type == 'cpu':
        return (dict(pts3d=host_tensor((P, H, W, 3)), conf=host_tensor((P, H, W))), dict(pts3d_in_other_view=host_tensor((P, H, W, 3)), conf=host_tensor((P, H, W))))
    kw = dict(dtype=torch.float32, device=out_device)
    return (dict(pts3d=torch.empty((P, H, W, 3), **kw), conf=torch.empty((P, H, W), **kw)),
            dict(pts3d_in_other_view=torch.empty((P, H, W, 3), **kw), conf=torch.empty((P, H, W), **kw)))


_SIDE_STREAMS = {}


def _side_stream(device, role):
    """The side stream of `role` on `device`, created once per process. A stream per inference() call meant an HSA queue created and -- whenever the garbage
    collector got to the previous call's -- destroyed per call: sporadic 30-100 ms stalls at arbitrary places of a one-pair call on the MI355X box
    (profiles/r05_y/cat_probe.log), 54 ms per call on average against 10 ms of GPU work."""
    device = torch.device(device)
    key = (device.index if device.index is not None else torch.cuda.current_device(), role)
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
    return _SIDE_STREAMS[key]


class _PredictionSink:
    """Where the per-batch predictions go. Device outputs: a plain copy. Host outputs (the reference's format, inference.py:68): the
    D2H copies of batch k are issued on a side stream AFTER batch k + 1 has been enqueued, so the blocking copy into the (pageable,
    pre-touched) result tensors overlaps the next batch's compute. Measured on the MI355X box (profiles/r02_*/host_probe.log): a D2H
    into touched pageable memory runs at PCIe speed (75 MB in 2 ms), into torch's pinned memory at 2.4 GB/s -- so no pinned staging."""

    KEYS = (('pred1', 'pts3d'), ('pred1', 'conf'), ('pred2', 'pts3d_in_other_view'), ('pred2', 'conf'))

    def __init__(self, n_pairs, H, W, out_device, compute_device, outputs=None):
        self.pred1, self.pred2 = outputs if outputs is not None else _alloc_outputs(n_pairs, H, W, out_device)
        self.out = dict(pred1=self.pred1, pred2=self.pred2)
        self.host = torch.device(out_device).type == 'cpu' and torch.device(compute_device).type == 'cuda'
        self.compute_device = torch.device(compute_device)
        self.stash = None
        if self.host:
            self.stream = _side_stream(compute_device, 'predictions')

    def _flush(self):
        if self.stash is None:
            return
        ev, i, j, src = self.stash
        self.stash = None
        self.stream.wait_event(ev)
        with torch.cuda.stream(self.stream):
            for (a, b), t in src.items():
                self.out[a][b][i:j].copy_(t)            # blocking for the host, concurrent with the main stream's kernels
                t.record_stream(self.stream)

    def put(self, i, j, p1, p2):
        src = {('pred1', 'pts3d'): p1['pts3d'], ('pred1', 'conf'): p1['conf'], ('pred2', 'pts3d_in_other_view'): p2['pts3d_in_other_view'],
               ('pred2', 'conf'): p2['conf']}
        if not self.host:
            for (a, b), t in src.items():
                self.out[a][b][i:j].copy_(t, non_blocking=True)
            return
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.compute_device))   # on the COMPUTE device's stream (not the process' current device): batch (i, j) is enqueued; now move the PREVIOUS batch while it runs
        prev, self.stash = self.stash, None
        if prev is not None:
            self.stash = prev
            self._flush()
        self.stash = (ev, i, j, src)

    def finish(self):
        if self.host:
            self._flush()
            self.stream.synchronize()
        elif self.compute_device.type == 'cuda':
            torch.cuda.synchronize(self.compute_device)
        return self.pred1, self.pred2


def _shared_images(pairs):
    """(list of distinct `img` tensors in first-use order, index of every pair's view-1 image, of every view-2 image) when the pair list
    shares image tensors (what make_pairs produces), else None."""
    uniq, index = {}, ([], [])
    for side in (0, 1):
        for p in pairs:
            t = p[side]['img']
            index[side].append(uniq.setdefault(id(t), (len(uniq), t))[0])
    if len(uniq) >= len(pairs) or not all(t.shape[0] == 1 for _, t in uniq.values()):
        return None
    return [t for _, t in sorted(uniq.values(), key=lambda x: x[0])], index[0], index[1]


def _collate_views(pairs, shared=None, device_stack=None, ready=None):
    """view1 / view2 dicts of the whole pair list in the reference's collated format (inference.py:68-72: tensors concatenated on the
    host, lists chained). When the pair list shares images, the two big `img` tensors (2 x len(pairs) images) are gathered ON THE GPU
    from the stack of distinct images (already uploaded for the forward) and come back in one D2H each: a host-side concatenation of
    600 pairs x 2 views ran 1.6 s per view on the GPU box (profiles/r02_*/host_probe.log), the device gather + copy ~0.1 s.
    `ready` (round 5): a CUDA event behind the upload of `device_stack`. The function is then safe to run on a host thread WHILE the forward
    loop runs (the two view tensors are 2.8 GB for 600 pairs at 512x384 -- a quarter second of page faults and PCIe that used to follow the
    last batch): its gathers and D2H copies go through a side stream of their own that waits for that event only."""
    if shared is None or device_stack is None:
        return collate_with_cat(list(pairs))
    light = [tuple({k: v for k, v in view.items() if k != 'img'} for view in p) for p in pairs]
    view1, view2 = collate_with_cat(light)
    dev = device_stack.device
    side = _side_stream(dev, 'views') if ready is not None else None

    def gather(index, chunk=128):      # in chunks: 2 x len(pairs) whole images never sit in HBM at once (2.8 GB for 600 pairs at 512x384)
        out = host_tensor((len(index),) + tuple(device_stack.shape[1:]), dtype=device_stack.dtype)     # zero-filled: the pages are touched before the copies need them
        if side is None:
            idx = torch.tensor(index, device=dev)
            for i in range(0, len(index), chunk):
                out[i:i + chunk].copy_(device_stack.index_select(0, idx[i:i + chunk]))
            return out
        with torch.cuda.device(dev), torch.cuda.stream(side):
            side.wait_event(ready)
            idx = torch.tensor(index, device=dev)
            for i in range(0, len(index), chunk):
                piece = device_stack.index_select(0, idx[i:i + chunk])
                out[i:i + chunk].copy_(piece)      # blocks this thread only; ordered on the side stream
            side.synchronize()
        return out
    view1['img'], view2['img'] = gather(shared[1]), gather(shared[2])
    return view1, view2


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
def inference_encode_once(pairs, model, device, batch_size=8, verbose=True, output_device='cpu', engine_batch=None):
    """Same return value as `inference` (bit-identical: every engine kernel is batch-position independent), but every distinct
    image goes through the ViT-L encoder ONCE: n encoder passes instead of 2 x len(pairs) -- 20 instead of 380 for the demo's
    complete symmetrised graph over 20 views, i.e. ~53 % fewer FLOPs end to end. Predictions stream to the host
    behind the compute (see _PredictionSink); the result tensors are allocated and touched on a host thread meanwhile."""
    imgs, order = {}, []
    for v1, v2 in pairs:
        for v in (v1, v2):
            k = int(v['idx'])
            if k not in imgs:
                imgs[k] = v['img']
                order.append(k)
    pos = {k: i for i, k in enumerate(order)}
    H, W = pairs[0][0]['img'].shape[-2:]
    outputs = _Background(lambda: _alloc_outputs(len(pairs), H, W, output_device))
    feats, dev_imgs = [], []
    batch_size = _engine_step(model, batch_size, engine_batch)
    enc_bs = max(2, 2 * batch_size)
    for i in tqdm.trange(0, len(order), enc_bs, disable=not verbose, desc='encode'):
        dev_imgs.append(upload_stack([imgs[k] for k in order[i:i + enc_bs]], device))
        feats.append(model.encode_images(dev_imgs[-1]))
    feats = torch.cat(feats, dim=0)
    i1h, i2h = [pos[int(a['idx'])] for a, _ in pairs], [pos[int(b['idx'])] for _, b in pairs]
    # the collated view images (what inference() returns next to the predictions) go to the host on a thread of their own while the pairs decode
    stack = torch.cat(dev_imgs, dim=0)
    views = None
    if stack.is_cuda:
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(stack.device))
        views = _Background(_collate_views, pairs, (None, i1h, i2h), stack, ready)
    sink = _PredictionSink(len(pairs), H, W, output_device, feats.device, outputs=outputs.join())
    i1, i2 = torch.tensor(i1h, device=feats.device), torch.tensor(i2h, device=feats.device)
    for i in tqdm.trange(0, len(pairs), batch_size, disable=not verbose, desc='decode'):
        j = min(i + batch_size, len(pairs))
        p1, p2 = model.decode_pairs(feats.index_select(0, torch.cat((i1[i:j], i2[i:j]))), H, W)
        sink.put(i, j, p1, p2)
    pred1, pred2 = sink.finish()
    view1, view2 = views.join() if views is not None else _collate_views(pairs, shared=(None, i1h, i2h), device_stack=stack)
    return dict(view1=view1, view2=view2, pred1=pred1, pred2=pred2, loss=None)


@torch.no_grad()
def inference(pairs, model, device, batch_size=8, verbose=True, encode_once=None, output_device='cpu', engine_batch=None):
    """Mirror of dust3r/inference.py:55-72. Extras (defaults keep the reference's behaviour): `encode_once` (None = automatic:
    encode every distinct image once when the pair list shares images), `output_device` ('cpu' like the reference's
    to_cpu; a CUDA device keeps the predictions in HBM for `global_aligner(output, device)`, which would upload them again) and
    `engine_batch`. `batch_size` is a LOWER bound on the pairs per engine call: the engine takes `model.engine_batch` pairs (32;
    DUST3R_AMD_ENGINE_BATCH) whatever smaller value the caller names, with bit-identical results; `engine_batch=n` pins the call size
    to exactly n pairs (use it to bound the workspace: ~0.8 GB per 512x384 pair in fp16x3)."""
    from .utils.device import fit_host_threads_once
    fit_host_threads_once()
    if verbose:
        print(f'>> Inference with model on {len(pairs)} image pairs')
    multiple_shapes = not check_if_same_size(pairs)
    if encode_once is None:
        encode_once = True
    if encode_once and not multiple_shapes and _encode_once_ok(pairs, model):
        return inference_encode_once(pairs, model, device, batch_size=batch_size, verbose=verbose, output_device=output_device, engine_batch=engine_batch)
    if multiple_shapes:
        # dust3r/inference.py:60-68 falls back to one pair per call. Here the pairs are grouped by their (view 1, view 2) image sizes,
        # every group runs `engine_batch` pairs per call, and the per-pair rows go back to their positions in the list: same structure
        # (lists, one entry per pair), bit-identical values.
        step = _engine_step(model, 1, engine_batch)
        groups = {}
        for k, (v1, v2) in enumerate(pairs):
            groups.setdefault((tuple(v1['img'].shape[-2:]), tuple(v2['img'].shape[-2:])), []).append(k)
        result = [None] * len(pairs)
        bar = tqdm.tqdm(total=len(pairs), disable=not verbose)
        for members in groups.values():
            for i in range(0, len(members), step):
                chunk = members[i:i + step]
                res = loss_of_one_batch(collate_with_cat([pairs[k] for k in chunk]), model, None, device)
                res = to_cpu(res) if str(output_device) == 'cpu' else res
                for r, k in enumerate(chunk):
                    result[k] = _rows_of(res, r)
                bar.update(len(chunk))
        bar.close()
        return collate_with_cat(result, lists=True)
    H, W = pairs[0][0]['img'].shape[-2:]
    batch_size = _engine_step(model, batch_size, engine_batch)
    on_gpu = torch.device(device).type == 'cuda'
    shared = _shared_images(pairs) if on_gpu else None
    sink = _PredictionSink(len(pairs), H, W, output_device, device)
    views = None
    if shared is not None:       # the distinct images go up once; every batch is gathered on the device
        stack = upload_stack(shared[0], device)
        i1, i2 = torch.tensor(shared[1], device=stack.device), torch.tensor(shared[2], device=stack.device)
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(stack.device))
        views = _Background(_collate_views, pairs, shared, stack, ready)      # the collated view images leave for the host while the batches run
    for i in tqdm.trange(0, len(pairs), batch_size, disable=not verbose):
        j = min(i + batch_size, len(pairs))
        if shared is not None:
            batch = collate_with_cat([tuple({k: v for k, v in view.items() if k != 'img'} for view in p) for p in pairs[i:j]])
            batch[0]['img'], batch[1]['img'] = stack.index_select(0, i1[i:j]), stack.index_select(0, i2[i:j])
        else:
            batch = collate_with_cat(pairs[i:j])
        res = loss_of_one_batch(batch, model, None, device)
        sink.put(i, j, res['pred1'], res['pred2'])
    pred1, pred2 = sink.finish()
    view1, view2 = views.join() if views is not None else _collate_views(pairs, shared, stack if shared is not None else None)
    return dict(view1=view1, view2=view2, pred1=pred1, pred2=pred2, loss=None)
