"""oracle/ -- TEST INFRASTRUCTURE ONLY.

CPU restatement of the reference's hot path (naver/dust3r @ /root/reference):
  croco_ref/models/*   restated naver/croco modules (submodule absent -> PARITY UNPINNED)
  roma_ref.py          restated `roma` subset           (dependency absent -> PARITY UNPINNED)
  dust3r_ref.py        restated dust3r glue: model.py / heads / postprocess / inference
  aligner_ref.py       restated cloud_opt PointCloudOptimizer forward + Adam loop
  f8_ref.py            the arithmetic of the engine's fp16 + fp8 contraction in fp64 (no reference counterpart: pins the mode's kernels)
  ref_import.py        (build container only) imports the UNMODIFIED reference files from
                       /root/reference on top of the restated croco/roma shims, to pin the
                       restatements above and to generate tests/golden/* (make_golden.py)

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package. The product (dust3r_amd/) never imports it and fails loudly without its HIP library.
"""


def usable_cpus(cap=32):
    """CPU threads this process may actually use: min(affinity mask, cgroup v2/v1 CPU quota, cap). The GPU boxes
    report 256 logical CPUs to os.cpu_count() but run the container under a much smaller quota; 256 OpenMP threads on
    such a box spin against each other (measured: 481 s per oracle forward instead of ~10 s)."""
    import os
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        with open('/sys/fs/cgroup/cpu.max') as f:
            quota, period = f.read().split()[:2]
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period) + 0.5)))
    except Exception:
        try:
            with open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us') as f:
                q = int(f.read())
            with open('/sys/fs/cgroup/cpu/cpu.cfs_period_us') as f:
                p = int(f.read())
            if q > 0:
                n = min(n, max(1, int(q / p + 0.5)))
        except Exception:
            pass
    return max(1, min(n, cap))


def tune_threads(candidates=(8, 16, 32, 64, 128), verbose=False):
    """Pick torch's intra-op thread count by measurement (a ViT-sized fp32 matmul), among candidates not exceeding
    usable_cpus(cap=256), and set it. Returns the chosen count."""
    import time
    import torch
    lim = usable_cpus(cap=256)
    cands = sorted({c for c in candidates if c <= lim} | {min(lim, 8)})
    a, b = torch.randn(1536, 1024), torch.randn(1024, 4096)
    best, best_t = cands[0], float('inf')
    for n in cands:
        torch.set_num_threads(n)
        (a @ b).sum()
        t = time.perf_counter()
        for _ in range(4):
            (a @ b).sum()
        t = time.perf_counter() - t
        if verbose:
            print(f'[oracle] {n} threads: {t * 250:.1f} ms / matmul')
        if t < best_t * 0.9:          # prefer fewer threads unless clearly faster
            best, best_t = n, t
    torch.set_num_threads(best)
    return best
