"""Small shared pieces of the aligner's host side (names as in the reference `dust3r/cloud_opt/commons.py`): edge keys, image shapes from
the edge list, the confidence transform, the signed log / exp parameterisation of translations and the two learning-rate schedules.
(Edge scores and the distance functions live in the HIP kernels.)"""
import numpy as np
import torch


def edge_str(i, j):
    return f'{i}_{j}'


def get_imshapes(edges, pred_i, pred_j):
    n_imgs = max(max(e) for e in edges) + 1
    imshapes = [None] * n_imgs
    for e, (i, j) in enumerate(edges):
        shape_i, shape_j = tuple(pred_i[e].shape[0:2]), tuple(pred_j[e].shape[0:2])
        if imshapes[i]:
            assert imshapes[i] == shape_i, f'incorrect shape for image {i}'
        if imshapes[j]:
            assert imshapes[j] == shape_j, f'incorrect shape for image {j}'
        imshapes[i], imshapes[j] = shape_i, shape_j
    return imshapes


def get_conf_trf(mode):
    if mode == 'log':
        return lambda x: x.log()
    if mode == 'sqrt':
        return lambda x: x.sqrt()
    if mode == 'm1':
        return lambda x: x - 1
    if mode in ('id', 'none'):
        return lambda x: x
    raise ValueError(f'bad mode for {mode=}')


def signed_log1p(x):
    return torch.sign(x) * torch.log1p(torch.abs(x))


def signed_expm1(x):
    return torch.sign(x) * torch.expm1(torch.abs(x))


def cosine_schedule(t, lr_start, lr_end):
    assert 0 <= t <= 1
    return lr_end + (lr_start - lr_end) * (1 + np.cos(t * np.pi)) / 2


def linear_schedule(t, lr_start, lr_end):
    assert 0 <= t <= 1
    return lr_start + (lr_end - lr_start) * t
