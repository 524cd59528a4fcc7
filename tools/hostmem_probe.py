"""Host memory for the results of inference(): what allocating + first-touching 3.8 GB of result tensors costs on the box, by allocation route
(torch.zeros = malloc + memset; anonymous mmap with MADV_HUGEPAGE; MAP_POPULATE), and how fast a device -> host copy lands in each.
Usage: python tools/hostmem_probe.py [GB=3.8]"""
import mmap
import sys
import time

import torch


def main():
    gb = float(sys.argv[1]) if len(sys.argv) > 1 else 3.8
    nbytes = int(gb * 2**30) // (1 << 21) * (1 << 21)
    for f in ('enabled', 'defrag', 'shmem_enabled'):
        try:
            print(f'transparent_hugepage/{f}:', open(f'/sys/kernel/mm/transparent_hugepage/{f}').read().strip())
        except OSError as e:
            print(f'transparent_hugepage/{f}: {e}')
    dev = torch.device('cuda:0')
    src = torch.rand(75 * 2**20 // 4, device=dev)          # one batch of predictions: ~75 MB per tensor kind
    torch.cuda.synchronize()

    def d2h(dst):
        n = src.numel()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for i in range(0, dst.numel() - n + 1, n):
            dst[i:i + n].copy_(src)
        torch.cuda.synchronize()
        return time.perf_counter() - t

    def route_zeros():
        return torch.zeros(nbytes // 4, dtype=torch.float32), None

    def route_mmap(advise, populate):
        flags = mmap.MAP_PRIVATE | mmap.MAP_ANONYMOUS | (getattr(mmap, 'MAP_POPULATE', 0) if populate else 0)
        mm = mmap.mmap(-1, nbytes, flags=flags)
        if advise:
            mm.madvise(mmap.MADV_HUGEPAGE)
        t = torch.frombuffer(mm, dtype=torch.float32)
        if not populate:
            t.zero_()
        return t, mm

    for rep in range(2):
        for name, fn in (('torch.zeros', route_zeros), ('mmap + zero_', lambda: route_mmap(False, False)), ('mmap + MADV_HUGEPAGE + zero_', lambda: route_mmap(True, False)),
                         ('mmap MAP_POPULATE', lambda: route_mmap(False, True)), ('mmap MADV_HUGEPAGE, untouched', None)):
            t0 = time.perf_counter()
            if fn is None:
                mm = mmap.mmap(-1, nbytes, flags=mmap.MAP_PRIVATE | mmap.MAP_ANONYMOUS)
                mm.madvise(mmap.MADV_HUGEPAGE)
                t = torch.frombuffer(mm, dtype=torch.float32)
            else:
                t, mm = fn()
            t1 = time.perf_counter()
            c1 = d2h(t)
            c2 = d2h(t)
            print(f'  rep {rep} {name:32s}: allocate + touch {1e3 * (t1 - t0):7.1f} ms | first D2H pass {1e3 * c1:7.1f} ms ({nbytes / c1 / 1e9:5.1f} GB/s) | second {1e3 * c2:7.1f} ms ({nbytes / c2 / 1e9:5.1f} GB/s)', flush=True)
            del t, mm


def upload_probe():
    """The other direction: 100 separately allocated (1, 3, 384, 512) images (what load_images returns) -> one device stack."""
    dev = torch.device('cuda:0')
    n, shp = 100, (1, 3, 384, 512)
    nb = n * 3 * 384 * 512 * 4

    def fresh():
        return [torch.rand(shp) for _ in range(n)]

    def hp_buffer():
        mm = mmap.mmap(-1, (nb + (1 << 21) - 1) // (1 << 21) * (1 << 21), flags=mmap.MAP_PRIVATE | mmap.MAP_ANONYMOUS)
        mm.madvise(mmap.MADV_HUGEPAGE)
        return torch.frombuffer(mm, dtype=torch.float32, count=nb // 4).view(n, *shp[1:])

    def route_cat(imgs):
        return torch.cat(imgs, 0).to(dev)

    def route_each(imgs):
        out = torch.empty((n,) + shp[1:], device=dev)
        for i, t in enumerate(imgs):
            out[i:i + 1].copy_(t, non_blocking=True)
        return out

    def route_hp(imgs):
        buf = hp_buffer()
        torch.cat(imgs, 0, out=buf)
        return buf.to(dev)

    pinned = torch.empty((n,) + shp[1:], pin_memory=True)

    def route_pinned(imgs):
        torch.cat(imgs, 0, out=pinned)
        return pinned.to(dev, non_blocking=True)

    def route_chunks(imgs, c=16):
        out = torch.empty((n,) + shp[1:], device=dev)
        for i in range(0, n, c):
            out[i:i + c].copy_(torch.cat(imgs[i:i + c], 0), non_blocking=True)
        return out

    for rep in range(3):
        for name, fn in (('torch.cat(...).to(dev)', route_cat), ('one copy per image', route_each), ('cat into huge pages, .to(dev)', route_hp),
                         ('cat into a cached pinned buffer', route_pinned), ('cat + copy in chunks of 16', route_chunks)):
            imgs = fresh()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = fn(imgs)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            print(f'  rep {rep} upload {name:34s}: {1e3 * dt:7.1f} ms ({nb / dt / 1e9:5.1f} GB/s)', flush=True)
            del out, imgs


def pieces_probe(gb=1.9):
    """Device -> host into FRESH result memory, by piece size of the copies and by allocation route (the runtime moves small pageable copies through its
    pinned staging buffers and large ones by pinning the destination on the fly)."""
    dev = torch.device('cuda:0')
    nbytes = int(gb * 2**30) // (1 << 21) * (1 << 21)
    src = torch.rand(nbytes // 4, device=dev)
    torch.cuda.synchronize()

    def alloc(route):
        if route == 'empty':
            return torch.empty(nbytes // 4), None
        if route == 'zeros':
            return torch.zeros(nbytes // 4), None
        mm = mmap.mmap(-1, nbytes, flags=mmap.MAP_PRIVATE | mmap.MAP_ANONYMOUS)
        mm.madvise(mmap.MADV_HUGEPAGE)
        t = torch.frombuffer(mm, dtype=torch.float32)
        if route == 'huge+touch':
            t.zero_()
        return t, mm

    for rep in range(2):
        for route in ('empty', 'zeros', 'huge', 'huge+touch'):
            for mb in (0.75, 2.25, 9, 72):
                n = int(mb * 2**20) // 4
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                dst, mm = alloc(route)
                t1 = time.perf_counter()
                for i in range(0, dst.numel() - n + 1, n):
                    dst[i:i + n].copy_(src[i:i + n], non_blocking=True)
                torch.cuda.synchronize()
                t2 = time.perf_counter()
                print(f'  rep {rep} {route:10s} pieces of {mb:5.2f} MB: allocate {1e3 * (t1 - t0):6.1f} ms + copy {1e3 * (t2 - t1):6.1f} ms ({nbytes / (t2 - t1) / 1e9:5.1f} GB/s) = {1e3 * (t2 - t0):6.1f} ms', flush=True)
                del dst, mm


if __name__ == '__main__':
    if '--pieces' in sys.argv:
        pieces_probe()
        sys.exit(0)
    if '--upload' in sys.argv:
        sys.argv.remove('--upload')
        upload_probe()
        sys.exit(0)
    main()
