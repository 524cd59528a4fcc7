"""Image loading for the engine's input boundary (the reference's `dust3r/utils/image.py:64-128` `load_images`, `rgb`), without
torchvision / cv2: PIL decodes and resamples (the reference's filter choice: LANCZOS when shrinking, BICUBIC otherwise), everything
else is integer geometry and one normalisation pass.

Split in two so that the geometry is testable on its own and the pixels can stay out of Python:
  fit_geometry(W1, H1, size, ...)  -> what `load_images` does to an image of that size: the resample size and the crop box
  load_images(...)                 -> list of dict(img=(1,3,H,W) fp32 in [-1,1], true_shape=int32 [[H, W]], idx=int, instance=str)
`device=` makes the normalisation ((x / 255 - 0.5) / 0.5, the reference's ToTensor + Normalize) run on that device from the uint8
pixels, so a CUDA caller uploads 1 byte per sample instead of 4; the values are the same fp32 values.
"""
import os

import numpy as np
import PIL.Image
import torch
from PIL.ImageOps import exif_transpose

IMAGE_EXTENSIONS = ('.jpg', '.jpeg', '.png')


def fit_geometry(W1, H1, size, square_ok=False, patch_size=16):
    """(resampled (W, H), crop box (left, top, right, bottom) in the resampled image) for a W1 x H1 source.
    size == 224: short side to 224 (long side scaled accordingly), centred square crop of side 2 * (min(W, H) // 2).
    otherwise  : long side to `size`; centred crop whose half extents are the largest multiples of patch_size / 2 that fit
                 around the integer centre; an exactly square result becomes 4:3 (half height = 3/4 half width) unless square_ok."""
    def scaled(long_edge):
        S = max(W1, H1)
        return tuple(int(round(x * long_edge / S)) for x in (W1, H1))
    if size == 224:
        W, H = scaled(round(size * max(W1 / H1, H1 / W1)))
        cx, cy = W // 2, H // 2
        half = min(cx, cy)
        return (W, H), (cx - half, cy - half, cx + half, cy + half)
    W, H = scaled(size)
    cx, cy = W // 2, H // 2
    halfw = ((2 * cx) // patch_size) * patch_size / 2
    halfh = ((2 * cy) // patch_size) * patch_size / 2
    if W == H and not square_ok:
        halfh = 3 * halfw / 4
    return (W, H), (cx - halfw, cy - halfh, cx + halfw, cy + halfh)


def normalize_pixels(u8_hwc, device=None):
    """uint8 (H, W, 3) -> fp32 (1, 3, H, W) in [-1, 1]: (x / 255 - 0.5) / 0.5, evaluated on `device` (default: where the pixels are)."""
    t = torch.from_numpy(np.array(u8_hwc, dtype=np.uint8, order='C'))    # a copy: PIL hands out read-only buffers
    if device is not None:
        t = t.to(device, non_blocking=True)
    t = t.permute(2, 0, 1).float().div(255)
    return ((t - 0.5) / 0.5)[None]


def rgb(ftensor, true_shape=None):
    """Back to displayable RGB in [0, 1], HWC (image.py:45-61): uint8 is divided by 255, floats are un-normalised."""
    if isinstance(ftensor, list):
        return [rgb(x, true_shape=true_shape) for x in ftensor]
    a = ftensor.detach().cpu().numpy() if isinstance(ftensor, torch.Tensor) else ftensor
    if a.ndim == 3 and a.shape[0] == 3:
        a = a.transpose(1, 2, 0)
    elif a.ndim == 4 and a.shape[1] == 3:
        a = a.transpose(0, 2, 3, 1)
    if true_shape is not None:
        a = a[:true_shape[0], :true_shape[1]]
    a = np.float32(a) / 255 if a.dtype == np.uint8 else a * 0.5 + 0.5
    return a.clip(min=0, max=1)


def load_images(folder_or_list, size, square_ok=False, verbose=True, patch_size=16, device=None):
    if isinstance(folder_or_list, str):
        if verbose:
            print(f'>> Loading images from {folder_or_list}')
        root, names = folder_or_list, sorted(os.listdir(folder_or_list))
    elif isinstance(folder_or_list, list):
        if verbose:
            print(f'>> Loading a list of {len(folder_or_list)} images')
        root, names = '', folder_or_list
    else:
        raise ValueError(f'bad {folder_or_list=} ({type(folder_or_list)})')
    names = [name for name in names if name.lower().endswith(IMAGE_EXTENSIONS)]

    def decode(name):
        """One file -> (cropped uint8 pixels, source size). PIL releases the GIL while it decodes and resamples, so the files of a folder are
        prepared on a thread pool (DUST3R_AMD_LOAD_THREADS, default min(16, usable cores)): same PIL calls per image, same pixels, same order."""
        pil = exif_transpose(PIL.Image.open(os.path.join(root, name))).convert('RGB')
        W1, H1 = pil.size
        new_size, box = fit_geometry(W1, H1, size, square_ok=square_ok, patch_size=patch_size)
        # the filter is chosen on the source's long edge against the long-edge TARGET of the resize call (not the rounded result)
        target = round(size * max(W1 / H1, H1 / W1)) if size == 224 else size
        pil = pil.resize(new_size, PIL.Image.LANCZOS if max(W1, H1) > target else PIL.Image.BICUBIC).crop(box)
        return np.asarray(pil, dtype=np.uint8), (W1, H1)

    from .device import usable_cpus
    threads = max(1, min(int(os.environ.get('DUST3R_AMD_LOAD_THREADS', 16)), usable_cpus(), len(names)))
    if threads > 1:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=threads) as pool:
            decoded = list(pool.map(decode, names))
    else:
        decoded = [decode(name) for name in names]
    views = []
    for name, (pixels, (W1, H1)) in zip(names, decoded):
        H2, W2 = pixels.shape[:2]
        if verbose:
            print(f' - adding {name} with resolution {W1}x{H1} --> {W2}x{H2}')
        views.append(dict(img=normalize_pixels(pixels, device), true_shape=np.int32([[H2, W2]]), idx=len(views), instance=str(len(views))))
    assert views, 'no images foud at ' + root
    if verbose:
        print(f' (Found {len(views)} images)')
    return views
