"""`global_aligner(dust3r_output, device, mode, **optim_kw)` -- the entry point of the alignment stage (reference
`dust3r/cloud_opt/__init__.py:14-33`), with the reference's mode enum."""
from enum import Enum

from .optimizer import PointCloudOptimizer
from .pair_viewer import PairViewer


class GlobalAlignerMode(Enum):
    PointCloudOptimizer = "PointCloudOptimizer"
    ModularPointCloudOptimizer = "ModularPointCloudOptimizer"
    PairViewer = "PairViewer"


_SCENES = {GlobalAlignerMode.PointCloudOptimizer: PointCloudOptimizer, GlobalAlignerMode.PairViewer: PairViewer}


def global_aligner(dust3r_output, device, mode=GlobalAlignerMode.PointCloudOptimizer, **optim_kw):
    if mode == GlobalAlignerMode.ModularPointCloudOptimizer:
        raise NotImplementedError('ModularPointCloudOptimizer (the slow per-edge variant, unused by the demo) is out of scope: '
                                  'use GlobalAlignerMode.PointCloudOptimizer')
    if mode not in _SCENES:
        raise NotImplementedError(f'Unknown mode {mode}')
    inputs = [dust3r_output[k] for k in ('view1', 'view2', 'pred1', 'pred2')]
    return _SCENES[mode](*inputs, **optim_kw).to(device)
