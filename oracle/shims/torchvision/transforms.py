import numpy as np
import torch


class Compose:
    def __init__(self, fns):
        self.fns = fns

    def __call__(self, x):
        for f in self.fns:
            x = f(x)
        return x


class ToTensor:
    def __call__(self, pil_img):
        arr = np.asarray(pil_img, dtype=np.uint8)
        if arr.ndim == 2:
            arr = arr[:, :, None]
        return torch.from_numpy(arr.copy()).permute(2, 0, 1).float().div(255)


class Normalize:
    def __init__(self, mean, std):
        self.mean = torch.tensor(mean).view(-1, 1, 1)
        self.std = torch.tensor(std).view(-1, 1, 1)

    def __call__(self, t):
        return (t - self.mean) / self.std
