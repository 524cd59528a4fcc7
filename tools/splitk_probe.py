"""On-GPU probe (not a test): the 1 / 2 / 4-pair forward with the split-K rule of gemm.hip (launch_gemm) off and on, interleaved in one process
(the same clocks for every arm). Usage: python tools/splitk_probe.py ["TILE,BLOCKS,MINSTEPS" ...]   -- extra arms set the probe knobs D3R_SK_TILE /
D3R_SK_BLOCKS / D3R_SK_MINSTEPS (probe builds: D3R_PROBES=1 python -m dust3r_amd.build)."""
import os
import sys
import time

import torch

sys.path.insert(0, '.')
from bench import build_model  # noqa: E402
from dust3r_amd.synthetic import synthetic_views  # noqa: E402

dev = torch.device('cuda:0')


def ms_per_call(model, a, b, reps):
    for _ in range(3):
        model(a, b)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        model(a, b)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e3


def main():
    model = build_model('fp16x3', dev)
    v1, v2 = synthetic_views(8, 384, 512, seed=0, device=dev)
    for nb, reps in ((1, 30), (2, 20), (4, 10)):
        sub = lambda v: {k: x[:nb] for k, x in v.items()}   # noqa: E731
        a, b = sub(v1), sub(v2)
        cells = []
        arms = [None, ''] + sys.argv[1:]
        for rnd in range(2):
            for arm in arms:
                model.set_split_k(arm is not None)
                for k in ('D3R_SK_TILE', 'D3R_SK_BLOCKS', 'D3R_SK_MINSTEPS'):
                    os.environ.pop(k, None)
                if arm:
                    t, bl, ms = arm.split(',')
                    os.environ.update(D3R_SK_TILE=t, D3R_SK_BLOCKS=bl, D3R_SK_MINSTEPS=ms)
                cells.append(f'{"off" if arm is None else (arm or "on")} {ms_per_call(model, a, b, reps):.3f}')
        print(f'pairs {nb}: ' + ' | '.join(cells), flush=True)
    model.set_split_k(True)


if __name__ == '__main__':
    main()
