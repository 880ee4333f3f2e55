"""CLIP validation transform: resize (bicubic, antialias) so the shorter side is ``image_size`` ->
centre crop -> RGB -> [0, 1] -> normalise (/root/reference/src/vit_prisma/transforms/model_transforms.py:9-20).

``get_clip_val_transforms()``  the reference's CPU pipeline on PIL images (what torchvision's Compose does there:
                               ``Resize`` on a PIL image IS ``Image.resize(..., BICUBIC)``), returning a ``[3, S, S]`` tensor.
``GpuClipTransform``           the same pipeline on the GPU for batches of decoded uint8 images, so that the CPU
                               DataLoader stops being the bottleneck once the ViT runs at >100 k images/s: antialiased
                               bicubic ``F.interpolate`` on device (agrees with PIL's filter up to PIL's own uint8
                               rounding, see the test), crop, normalise, cast to the model dtype -- one H2D copy of the raw
                               uint8 pixels per batch instead of float tensors.
"""
from __future__ import annotations

from typing import Callable, Iterable, List, Sequence, Union

import numpy as np
import torch
import torch.nn.functional as F

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def _resized_size(w: int, h: int, size: int):
    """torchvision.transforms.Resize(int): the shorter side becomes ``size``, the other int(size * long / short)."""
    if w <= h:
        return size, int(size * h / w)
    return int(size * w / h), size


def get_clip_val_transforms(image_size: int = 224, mean: Sequence[float] = CLIP_MEAN, std: Sequence[float] = CLIP_STD) -> Callable:
    m = torch.tensor(mean, dtype=torch.float32)[:, None, None]
    s = torch.tensor(std, dtype=torch.float32)[:, None, None]

    def transform(image):
        from PIL import Image
        w, h = image.size
        nw, nh = _resized_size(w, h, image_size)
        image = image.resize((nw, nh), Image.BICUBIC)
        left, top = int(round((nw - image_size) / 2.0)), int(round((nh - image_size) / 2.0))
        image = image.crop((left, top, left + image_size, top + image_size)).convert("RGB")
        x = torch.from_numpy(np.asarray(image, dtype=np.uint8).copy()).permute(2, 0, 1).float().div_(255.0)
        return (x - m) / s

    return transform


class GpuClipTransform:
    def __init__(self, image_size: int = 224, mean: Sequence[float] = CLIP_MEAN, std: Sequence[float] = CLIP_STD,
                 device: Union[str, torch.device] = "cuda", dtype: torch.dtype = torch.float32):
        self.size, self.device, self.dtype = image_size, torch.device(device), dtype
        self.mean = torch.tensor(mean, dtype=torch.float32, device=self.device)[None, :, None, None]
        self.std = torch.tensor(std, dtype=torch.float32, device=self.device)[None, :, None, None]

    def one(self, img_u8_hwc: torch.Tensor) -> torch.Tensor:
        """``[H, W, 3]`` uint8 (host or device) -> ``[1, 3, S, S]`` normalised, on the device."""
        x = img_u8_hwc.to(self.device, non_blocking=True).permute(2, 0, 1)[None].float()
        h, w = x.shape[-2:]
        nw, nh = _resized_size(w, h, self.size)
        x = F.interpolate(x, size=(nh, nw), mode="bicubic", antialias=True, align_corners=False)
        x = x.round_().clamp_(0.0, 255.0)                          # PIL resamples into uint8
        top, left = int(round((nh - self.size) / 2.0)), int(round((nw - self.size) / 2.0))
        x = x[:, :, top:top + self.size, left:left + self.size]
        return ((x / 255.0 - self.mean) / self.std).to(self.dtype)

    def __call__(self, images: Union[torch.Tensor, Iterable[torch.Tensor]]) -> torch.Tensor:
        """A list of ``[H, W, 3]`` uint8 tensors of any sizes, or one ``[B, H, W, 3]`` batch -> ``[B, 3, S, S]``."""
        if isinstance(images, torch.Tensor) and images.ndim == 4:
            x = images.to(self.device, non_blocking=True).permute(0, 3, 1, 2).float()
            h, w = x.shape[-2:]
            nw, nh = _resized_size(w, h, self.size)
            x = F.interpolate(x, size=(nh, nw), mode="bicubic", antialias=True, align_corners=False).round_().clamp_(0.0, 255.0)
            top, left = int(round((nh - self.size) / 2.0)), int(round((nw - self.size) / 2.0))
            x = x[:, :, top:top + self.size, left:left + self.size]
            return ((x / 255.0 - self.mean) / self.std).to(self.dtype)
        outs: List[torch.Tensor] = [self.one(im) for im in images]
        return torch.cat(outs, dim=0)
