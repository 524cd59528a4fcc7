"""Batch-shape glue -- mirror of the parts of the reference `dust3r/utils/misc.py:32-96` that the
inference path uses."""
import torch


def is_symmetrized(gt1, gt2):
    """True when the batch is [(a,b),(b,a),(c,d),(d,c),...] (instances compared pairwise)."""
    x, y = gt1['instance'], gt2['instance']
    if len(x) == len(y) and len(x) == 1:
        return False
    ok = True
    for i in range(0, len(x), 2):
        ok = ok and (x[i] == y[i + 1]) and (x[i + 1] == y[i])
    return ok


def flip(tensor):
    """tensor[0::2] <=> tensor[1::2]"""
    return torch.stack((tensor[1::2], tensor[0::2]), dim=1).flatten(0, 1)


def interleave(tensor1, tensor2):
    res1 = torch.stack((tensor1, tensor2), dim=1).flatten(0, 1)
    res2 = torch.stack((tensor2, tensor1), dim=1).flatten(0, 1)
    return res1, res2


def transposed(dic):
    return {k: v.swapaxes(1, 2) for k, v in dic.items()}
