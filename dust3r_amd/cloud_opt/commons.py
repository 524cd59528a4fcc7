"""Small shared pieces of the aligner's host side (names as in the reference `dust3r/cloud_opt/commons.py`): edge keys, image shapes from
the edge list, the confidence transform, the signed log / exp parameterisation of translations and the two learning-rate schedules.
(Edge scores and the distance functions live in the HIP kernels.)"""
import math

import torch


def edge_str(i, j):
    return f'{i}_{j}'


def get_imshapes(edges, pred_i, pred_j):
    """(H, W) of every image, read off the pairwise predictions; an image must have one shape in all the edges that show it."""
    shapes = {}
    for e, (i, j) in enumerate(edges):
        for img, pred in ((i, pred_i[e]), (j, pred_j[e])):
            hw = tuple(pred.shape[0:2])
            assert shapes.setdefault(img, hw) == hw, f'incorrect shape for image {img}'
    return [shapes.get(k) for k in range(max(max(e) for e in edges) + 1)]


_CONF_TRANSFORMS = {'log': torch.log, 'sqrt': torch.sqrt, 'm1': lambda x: x - 1, 'id': lambda x: x, 'none': lambda x: x}


def get_conf_trf(mode):
    if mode not in _CONF_TRANSFORMS:
        raise ValueError(f'bad mode for {mode=}')
    return _CONF_TRANSFORMS[mode]


def signed_log1p(x):
    return torch.sign(x) * torch.log1p(torch.abs(x))


def signed_expm1(x):
    return torch.sign(x) * torch.expm1(torch.abs(x))


def cosine_schedule(t, lr_start, lr_end):
    assert 0 <= t <= 1
    return lr_end + (lr_start - lr_end) * (1 + math.cos(t * math.pi)) / 2


def linear_schedule(t, lr_start, lr_end):
    assert 0 <= t <= 1
    return lr_start + (lr_end - lr_start) * t
