"""Where does inference()'s host side spend its time on this box? (not a test) python tools/host_probe.py"""
import sys
import time

import torch

sys.path.insert(0, '.')
dev = torch.device('cuda:0')
P, H, W = 600, 384, 512


def t(label, fn, reps=1):
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(reps):
        r = fn()
    torch.cuda.synchronize()
    dt = (time.time() - t0) / reps
    print(f'  {label:70s} {dt * 1e3:9.1f} ms')
    return r


print(f'torch threads {torch.get_num_threads()}')
a = t('pageable torch.empty (600,H,W,3) fp32 = 1.4 GB', lambda: torch.empty((P, H, W, 3)))
t('  first touch (fill_)', lambda: a.fill_(1.0))
t('  second touch (fill_)', lambda: a.fill_(2.0))
b = t('pinned torch.empty (600,H,W,3) = 1.4 GB', lambda: torch.empty((P, H, W, 3), pin_memory=True))
t('  first touch (fill_)', lambda: b.fill_(1.0))
g = torch.empty((32, H, W, 3), device=dev)
pin = torch.empty((32, H, W, 3), pin_memory=True)
t('D2H 32 pairs pts (75 MB) -> pinned', lambda: pin.copy_(g, non_blocking=True), reps=5)
t('D2H 32 pairs pts (75 MB) -> pageable slice (touched)', lambda: a[:32].copy_(g), reps=5)
t('host memcpy pinned -> pageable slice (75 MB)', lambda: a[32:64].copy_(pin), reps=5)
imgs = [torch.rand((1, 3, H, W)) for _ in range(100)]
stack = torch.cat(imgs)
idx = torch.randint(0, 100, (600,))
t('index_select 600 images from a 100-image stack (1.4 GB)', lambda: stack.index_select(0, idx))
t('torch.cat of 600 (1,3,H,W) tensors (1.4 GB)', lambda: torch.cat([imgs[i] for i in idx.tolist()]))
t('H2D 64 images pageable (151 MB)', lambda: torch.cat(imgs[:64]).to(dev), reps=3)
