"""Image loading -- mirror of the reference `dust3r/utils/image.py:45-128` (`rgb`, `load_images`)
without its torchvision / cv2 dependencies: PIL does the resize (LANCZOS when shrinking, BICUBIC when
enlarging, as the reference) and the [-1, 1] normalisation of `ImgNorm` is two tensor ops.
Output format (the engine's input boundary): a list of
  dict(img=(1,3,H,W) fp32 in [-1,1], true_shape=int32 [[H, W]], idx=int, instance=str)."""
import os

import numpy as np
import PIL.Image
import torch
from PIL.ImageOps import exif_transpose


def img_norm(pil_img):
    """ToTensor + Normalize((0.5,)*3, (0.5,)*3)  (image.py:23)."""
    arr = np.asarray(pil_img, dtype=np.uint8)
    t = torch.from_numpy(arr.copy()).permute(2, 0, 1).float().div(255)
    return (t - 0.5) / 0.5


def rgb(ftensor, true_shape=None):
    if isinstance(ftensor, list):
        return [rgb(x, true_shape=true_shape) for x in ftensor]
    if isinstance(ftensor, torch.Tensor):
        ftensor = ftensor.detach().cpu().numpy()
    if ftensor.ndim == 3 and ftensor.shape[0] == 3:
        ftensor = ftensor.transpose(1, 2, 0)
    elif ftensor.ndim == 4 and ftensor.shape[1] == 3:
        ftensor = ftensor.transpose(0, 2, 3, 1)
    if true_shape is not None:
        H, W = true_shape
        ftensor = ftensor[:H, :W]
    img = np.float32(ftensor) / 255 if ftensor.dtype == np.uint8 else (ftensor * 0.5) + 0.5
    return img.clip(min=0, max=1)


def _resize_pil_image(img, long_edge_size):
    S = max(img.size)
    interp = PIL.Image.LANCZOS if S > long_edge_size else PIL.Image.BICUBIC
    new_size = tuple(int(round(x * long_edge_size / S)) for x in img.size)
    return img.resize(new_size, interp)


def load_images(folder_or_list, size, square_ok=False, verbose=True, patch_size=16):
    if isinstance(folder_or_list, str):
        if verbose:
            print(f'>> Loading images from {folder_or_list}')
        root, folder_content = folder_or_list, sorted(os.listdir(folder_or_list))
    elif isinstance(folder_or_list, list):
        if verbose:
            print(f'>> Loading a list of {len(folder_or_list)} images')
        root, folder_content = '', folder_or_list
    else:
        raise ValueError(f'bad {folder_or_list=} ({type(folder_or_list)})')

    imgs = []
    for path in folder_content:
        if not path.lower().endswith(('.jpg', '.jpeg', '.png')):
            continue
        img = exif_transpose(PIL.Image.open(os.path.join(root, path))).convert('RGB')
        W1, H1 = img.size
        if size == 224:   # short side -> 224, then centre crop to a square
            img = _resize_pil_image(img, round(size * max(W1 / H1, H1 / W1)))
        else:             # long side -> size
            img = _resize_pil_image(img, size)
        W, H = img.size
        cx, cy = W // 2, H // 2
        if size == 224:
            half = min(cx, cy)
            img = img.crop((cx - half, cy - half, cx + half, cy + half))
        else:             # crop to multiples of the patch size; squares become 4:3 unless square_ok
            halfw = ((2 * cx) // patch_size) * patch_size / 2
            halfh = ((2 * cy) // patch_size) * patch_size / 2
            if not square_ok and W == H:
                halfh = 3 * halfw / 4
            img = img.crop((cx - halfw, cy - halfh, cx + halfw, cy + halfh))
        W2, H2 = img.size
        if verbose:
            print(f' - adding {path} with resolution {W1}x{H1} --> {W2}x{H2}')
        imgs.append(dict(img=img_norm(img)[None], true_shape=np.int32([img.size[::-1]]), idx=len(imgs), instance=str(len(imgs))))
    assert imgs, 'no images foud at ' + root
    if verbose:
        print(f' (Found {len(imgs)} images)')
    return imgs
