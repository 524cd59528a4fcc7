"""On-GPU probe (not a test): the engine forward at 1 / 2 / 3 / 4 / 8 pairs per call under several settings of environment switches that the library reads per launch,
interleaved in one process (same clocks for every arm). Usage: python tools/env_latency_ab.py "" "D3R_GEMM_T96=0" ["VAR=a VAR2=b" ...]     ("" = defaults)"""
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
    arms = sys.argv[1:] or ['']
    touched = sorted({kv.split('=')[0] for arm in arms for kv in arm.split()})
    model = build_model('fp16x3', dev)
    v1, v2 = synthetic_views(8, 384, 512, seed=0, device=dev)
    for nb, reps in ((1, 30), (2, 20), (3, 15), (4, 10), (8, 6)):
        sub = lambda v: {k: x[:nb] for k, x in v.items()}   # noqa: E731
        a, b = sub(v1), sub(v2)
        cells = []
        for rnd in range(3):
            for arm in arms:
                for k in touched:
                    os.environ.pop(k, None)
                for kv in arm.split():
                    k, v = kv.split('=')
                    os.environ[k] = v
                cells.append(f'[{arm or "default"}] {ms_per_call(model, a, b, reps):.3f}')
        print(f'pairs {nb}: ' + ' | '.join(cells), flush=True)


if __name__ == '__main__':
    main()
