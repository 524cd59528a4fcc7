"""Container plumbing of the inference path: moving nested view / prediction structures between devices and collating per-batch
results into the reference's return format (`dust3r/utils/device.py:11-76`: `to_cpu`, `to_numpy`, `collate_with_cat`)."""
import mmap

import numpy as np
import torch


def usable_cpus(cap=64):
    """CPU threads this process may actually use: min(affinity mask, cgroup CPU quota, cap). The MI355X boxes report 256 logical CPUs to
    os.cpu_count() and run the container under a quota of 16: torch then sizes its intra-op pool at 128 threads, and a host-side `torch.cat` of ONE
    2.4 MB image now and then takes 30-100 ms instead of 0.03 (128 OpenMP threads passing a barrier on 16 cores; profiles/r05_y/cat_probe.log)."""
    import os
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    for path_q, path_p in (('/sys/fs/cgroup/cpu.max', None), ('/sys/fs/cgroup/cpu/cpu.cfs_quota_us', '/sys/fs/cgroup/cpu/cpu.cfs_period_us')):
        try:
            with open(path_q) as f:
                fields = f.read().split()
            if path_p is None:
                if fields[0] == 'max':
                    break
                quota, period = int(fields[0]), int(fields[1])
            else:
                quota = int(fields[0])
                with open(path_p) as f:
                    period = int(f.read())
                if quota <= 0:
                    break
            n = min(n, max(1, int(quota / period + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, min(n, cap))


def fit_host_threads():
    """Cap torch's intra-op thread pool at usable_cpus() (never raises it; DUST3R_AMD_KEEP_TORCH_THREADS=1 leaves torch alone). Called lazily by the entry points
    of the path (fit_host_threads_once): every host-side tensor operation of the inference path (collation, result allocation, the aligner's initialisation) runs on that pool."""
    import os
    if os.environ.get('DUST3R_AMD_KEEP_TORCH_THREADS', '') not in ('', '0'):
        return torch.get_num_threads()
    n = usable_cpus()
    try:                                       # one process per GPU under torch.distributed.run: the ranks of a node share the quota
        n = max(1, n // max(1, int(os.environ.get('LOCAL_WORLD_SIZE', '1'))))
    except ValueError:
        pass
    if torch.get_num_threads() > n:
        import logging
        logging.getLogger('dust3r_amd').info('torch intra-op threads %d -> %d (the CPUs this process can use; DUST3R_AMD_KEEP_TORCH_THREADS=1 keeps torch\'s count)',
                                             torch.get_num_threads(), n)
        torch.set_num_threads(n)
    return torch.get_num_threads()


_fitted = False


def fit_host_threads_once():
    """fit_host_threads() on the first call of the process only -- from the entry points of the path (inference(), global_aligner()), NOT at import: importing the
    package leaves the host application's torch thread pool alone."""
    global _fitted
    if not _fitted:
        _fitted = True
        fit_host_threads()


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


# ---------------------------------------------------------------------------------------------------- host memory of the results, image upload
_HUGE = 1 << 21


def host_tensor(shape, dtype=torch.float32):
    """Zero-filled CPU tensor for results that a device -> host copy is about to fill (the 3.8 GB of predictions and the 2.8 GB of collated view
    images `inference()` returns for 600 pairs at 512x384). Large tensors come from an anonymous mapping marked MADV_HUGEPAGE -- this image's
    kernel has transparent huge pages in `madvise` mode -- and are touched here: measured on the MI355X box (tools/hostmem_probe.py,
    profiles/r05_y) allocate + first touch of 1.9 GB takes 7 ms instead of 130-200 ms (torch.zeros: 4 KiB page faults), and the first
    device -> host pass into it runs at 35-39 GB/s instead of 16 (the runtime pins the destination of a large pageable copy on the fly: 512x
    fewer pages to pin). Falls back to torch.zeros where the platform has no madvise / huge pages, and for small tensors."""
    shape = tuple(int(x) for x in shape)
    n = 1
    for x in shape:
        n *= x
    item = torch.empty((), dtype=dtype).element_size()
    nbytes = n * item
    if nbytes < 8 * _HUGE or not hasattr(mmap, 'MADV_HUGEPAGE'):
        return torch.zeros(shape, dtype=dtype)
    try:
        mm = mmap.mmap(-1, (nbytes + _HUGE - 1) // _HUGE * _HUGE, flags=mmap.MAP_PRIVATE | mmap.MAP_ANONYMOUS)
        mm.madvise(mmap.MADV_HUGEPAGE)
        t = torch.frombuffer(mm, dtype=dtype, count=n).view(shape)       # the tensor keeps the mapping alive (buffer protocol)
    except (OSError, ValueError, AttributeError):
        return torch.zeros(shape, dtype=dtype)
    return t.zero_()


def upload_stack(tensors, device, non_blocking=True):
    """A list of equally shaped CPU tensors (1, ...) -> one (n, ...) tensor on `device`, ONE COPY PER LIST ENTRY into a device tensor allocated here.
    `torch.cat(tensors).to(device)` -- a fresh 236 MB pageable region handed to one copy -- runs at 0.5-1 GB/s on the MI355X box (100 images of
    512x384: 230-470 ms); the runtime moves copies of a few MB through its pinned staging buffers at 25-38 GB/s (the same 100 images one by one:
    6-9 ms; tools/hostmem_probe.py --upload, profiles/r05_y). Anything that is not that shape of input takes the plain route."""
    device = torch.device(device)
    ok = (device.type == 'cuda' and len(tensors) > 0 and all(isinstance(t, torch.Tensor) and t.device.type == 'cpu' and t.dim() >= 1 and t.shape[0] == 1
                                                              and t.shape == tensors[0].shape and t.dtype == tensors[0].dtype for t in tensors))
    if not ok:
        return torch.cat(list(tensors), dim=0).to(device, non_blocking=non_blocking)
    out = torch.empty((len(tensors),) + tuple(tensors[0].shape[1:]), dtype=tensors[0].dtype, device=device)
    for i, t in enumerate(tensors):
        out[i:i + 1].copy_(t, non_blocking=non_blocking)
    return out


def upload_rows(t, device, piece_bytes=4 << 20):
    """One large CPU tensor -> `device`, copied in pieces of a few MB along dim 0 (see upload_stack: a multi-GB pageable region handed to ONE copy moves at
    0.5-1 GB/s on the MI355X box, pieces of 2-4 MB at 25-38 GB/s). Small tensors, tensors already on a device and non-CUDA targets take `.to(device)`."""
    device = torch.device(device)
    if not isinstance(t, torch.Tensor) or t.device.type != 'cpu' or device.type != 'cuda' or t.dim() == 0 or t.numel() * t.element_size() <= 2 * piece_bytes:
        return t.to(device) if isinstance(t, torch.Tensor) else t
    t = t.detach()
    if not t.is_contiguous():
        t = t.contiguous()
    row = max(t[0].numel() * t.element_size(), 1)
    step = max(int(piece_bytes // row), 1)
    out = torch.empty(t.shape, dtype=t.dtype, device=device)
    for i in range(0, t.shape[0], step):
        out[i:i + step].copy_(t[i:i + step], non_blocking=True)
    return out
