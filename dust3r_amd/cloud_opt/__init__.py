"""`global_aligner(dust3r_output, device, mode, **optim_kw)` -- the entry point of the alignment stage (reference
`dust3r/cloud_opt/__init__.py:14-33`), with the reference's mode enum."""
from enum import Enum

import torch

from ..utils.device import upload_rows
from .optimizer import PointCloudOptimizer
from .pair_viewer import PairViewer


class GlobalAlignerMode(Enum):
    PointCloudOptimizer = "PointCloudOptimizer"
    ModularPointCloudOptimizer = "ModularPointCloudOptimizer"
    PairViewer = "PairViewer"


_SCENES = {GlobalAlignerMode.PointCloudOptimizer: PointCloudOptimizer, GlobalAlignerMode.PairViewer: PairViewer}


def global_aligner(dust3r_output, device, mode=GlobalAlignerMode.PointCloudOptimizer, **optim_kw):
    from ..utils.device import fit_host_threads_once
    fit_host_threads_once()
    if mode == GlobalAlignerMode.ModularPointCloudOptimizer:
        raise NotImplementedError('ModularPointCloudOptimizer (the slow per-edge variant, unused by the demo) is out of scope: '
                                  'use GlobalAlignerMode.PointCloudOptimizer')
    if mode not in _SCENES:
        raise NotImplementedError(f'Unknown mode {mode}')
    inputs = [dust3r_output[k] for k in ('view1', 'view2', 'pred1', 'pred2')]
    # The reference builds the scene where the predictions are (inference() returns them on the CPU) and moves it: `Scene(...).to(device)`
    # (cloud_opt/__init__.py:29-31). Same result, other order: the big prediction tensors go up FIRST, in pieces of a few MB (utils/device.py:upload_rows --
    # 3.8 GB for 600 pairs; one `.to(device)` per tensor moves pageable memory at 0.5-1 GB/s on the MI355X box), and the stacks, per-image confidences
    # and log-weights are then formed on the GPU instead of on the host cores.
    if torch.device(device).type == 'cuda':
        inputs[2:] = [{k: (upload_rows(v, device) if isinstance(v, torch.Tensor) else v) for k, v in pred.items()} for pred in inputs[2:]]
    return _SCENES[mode](*inputs, **optim_kw).to(device)
