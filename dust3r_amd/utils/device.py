"""Container plumbing of the inference path: moving nested view / prediction structures between devices and collating per-batch
results into the reference's return format (`dust3r/utils/device.py:11-76`: `to_cpu`, `to_numpy`, `collate_with_cat`)."""
import numpy as np
import torch


def _map_leaves(x, fn):
    """Apply fn to every leaf of nested dicts / lists / tuples, keeping the container types."""
    if isinstance(x, dict):
        return {k: _map_leaves(v, fn) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return type(x)(_map_leaves(v, fn) for v in x)
    return fn(x)


def todevice(batch, device, callback=None, non_blocking=False):
    """Tensors (and numpy arrays, converted) inside `batch` go to `device`; device == 'numpy' converts tensors to arrays instead.
    Everything else (strings, ints, None) passes through."""
    if callback:
        batch = callback(batch)

    def leaf(x):
        if device == 'numpy':
            return x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else x
        if isinstance(x, np.ndarray):
            x = torch.from_numpy(x)
        return x.to(device, non_blocking=non_blocking) if torch.is_tensor(x) else x
    return _map_leaves(batch, leaf)


to_device = todevice


def to_numpy(x):
    return todevice(x, 'numpy')


def to_cpu(x):
    return todevice(x, 'cpu')


def collate_with_cat(items, lists=False):
    """A list of per-batch results (tuples / dicts / tensors / python lists, nested) -> one result of the same structure: tensors and
    arrays concatenated along dim 0 (or, with `lists`, their rows chained into one python list), python lists chained, scalars and
    strings kept as the list they came in. A dict argument is collated value by value."""
    if isinstance(items, dict):
        return {k: collate_with_cat(v, lists=lists) for k, v in items.items()}
    if not isinstance(items, (list, tuple)) or len(items) == 0:
        return items
    first, seq_type = items[0], type(items)
    if first is None:
        return None
    if isinstance(first, (bool, int, float, str)):
        return items
    if isinstance(first, tuple):
        return seq_type(collate_with_cat(list(member), lists=lists) for member in zip(*items))
    if isinstance(first, dict):
        return {k: collate_with_cat([it[k] for it in items], lists=lists) for k in first}
    if isinstance(first, (torch.Tensor, np.ndarray)):
        if lists:
            return [row for it in items for row in it]
        return torch.cat([torch.from_numpy(it) if isinstance(it, np.ndarray) else it for it in items])
    out = seq_type()
    for it in items:
        out = out + it
    return out
