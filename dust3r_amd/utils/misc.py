"""Batch-shape helpers of the inference path (the reference's `dust3r/utils/misc.py:32-64`): recognising a symmetrised batch and
building / swapping interleaved batches."""
import torch


def is_symmetrized(gt1, gt2):
    """True for a batch laid out [(a,b),(b,a),(c,d),(d,c),...]: every even position's pair is its successor's pair reversed."""
    x, y = gt1['instance'], gt2['instance']
    if len(x) == len(y) == 1:
        return False
    return all(x[i] == y[i + 1] and x[i + 1] == y[i] for i in range(0, len(x), 2))


def flip(tensor):
    """Swap the two members of every consecutive pair along dim 0."""
    return torch.stack((tensor[1::2], tensor[0::2]), dim=1).flatten(0, 1)


def interleave(tensor1, tensor2):
    """(a0,b0,a1,b1,...) and (b0,a0,b1,a1,...)."""
    both = torch.stack((tensor1, tensor2), dim=1)
    return both.flatten(0, 1), both.flip(1).flatten(0, 1)


def transposed(dic):
    return {k: v.swapaxes(1, 2) for k, v in dic.items()}
