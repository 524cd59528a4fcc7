"""Device / collation helpers -- mirror of the reference `dust3r/utils/device.py:11-76`."""
import numpy as np
import torch


def todevice(batch, device, callback=None, non_blocking=False):
    """Recursively move tensors inside dict / list / tuple containers. device may be 'numpy'."""
    if callback:
        batch = callback(batch)
    if isinstance(batch, dict):
        return {k: todevice(v, device) for k, v in batch.items()}
    if isinstance(batch, (tuple, list)):
        return type(batch)(todevice(x, device) for x in batch)
    x = batch
    if device == 'numpy':
        return x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else x
    if x is None:
        return x
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(x)
    if torch.is_tensor(x):
        x = x.to(device, non_blocking=non_blocking)
    return x


to_device = todevice


def to_numpy(x):
    return todevice(x, 'numpy')


def to_cpu(x):
    return todevice(x, 'cpu')


def listify(elems):
    return [x for e in elems for x in e]


def collate_with_cat(whatever, lists=False):
    """Concatenate a list of per-batch results: tensors are cat'ed (or chained into lists when
    `lists`), python lists are chained, dicts / tuples are collated member-wise."""
    if isinstance(whatever, dict):
        return {k: collate_with_cat(v, lists=lists) for k, v in whatever.items()}
    if isinstance(whatever, (tuple, list)):
        if len(whatever) == 0:
            return whatever
        elem, T = whatever[0], type(whatever)
        if elem is None:
            return None
        if isinstance(elem, (bool, float, int, str)):
            return whatever
        if isinstance(elem, tuple):
            return T(collate_with_cat(x, lists=lists) for x in zip(*whatever))
        if isinstance(elem, dict):
            return {k: collate_with_cat([e[k] for e in whatever], lists=lists) for k in elem}
        if isinstance(elem, torch.Tensor):
            return listify(whatever) if lists else torch.cat(whatever)
        if isinstance(elem, np.ndarray):
            return listify(whatever) if lists else torch.cat([torch.from_numpy(x) for x in whatever])
        return sum(whatever, T())
