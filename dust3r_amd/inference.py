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


@torch.no_grad()
def inference(pairs, model, device, batch_size=8, verbose=True):
    if verbose:
        print(f'>> Inference with model on {len(pairs)} image pairs')
    result = []
    multiple_shapes = not check_if_same_size(pairs)
    if multiple_shapes:
        batch_size = 1
    for i in tqdm.trange(0, len(pairs), batch_size, disable=not verbose):
        res = loss_of_one_batch(collate_with_cat(pairs[i:i + batch_size]), model, None, device)
        result.append(to_cpu(res))
    return collate_with_cat(result, lists=multiple_shapes)
