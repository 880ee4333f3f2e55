"""CLIP validation transform: resize (bicubic, antialias) so the shorter side is ``image_size`` ->
centre crop -> RGB -> [0, 1] -> normalise (/root/reference/src/vit_prisma/transforms/model_transforms.py:9-20).

``get_clip_val_transforms()``  the reference's CPU pipeline on PIL images (what torchvision's Compose does there:
                               ``Resize`` on a PIL image IS ``Image.resize(..., BICUBIC)``), returning a ``[3, S, S]`` tensor.
``GpuClipTransform``           the same pipeline on the GPU for batches of decoded uint8 images, so that the CPU
                               DataLoader stops being the bottleneck once the ViT runs at >100 k images/s -- one H2D copy
                               of the raw uint8 pixels per batch instead of float tensors.  On an MI355X with the library
                               built it is ONE hand-written kernel (``pv_clip_preprocess``, csrc/preprocess.hip) that
                               reproduces Pillow's fixed-point two-pass resampler exactly: bit-identical to the reference's
                               CPU pipeline.  Elsewhere (CPU device, no library): antialiased bicubic ``F.interpolate``,
                               which agrees with PIL's filter up to PIL's own uint8 rounding (<= 1 level).
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


def _pil_coeffs(in_size: int, out_size: int):
    """Tap tables of Pillow's bicubic resampler for one axis (Resample.c precompute_coeffs + normalize_coeffs_8bpc, the
    resize torchvision's ``Resize`` runs on PIL images): ``bounds [out, 2]`` = (first input index, tap count), ``taps
    [out, ksize]`` int32 with 22 fractional bits.  Same double-precision operations in the same order as the C code (the
    normaliser is a sequential sum), so the tables -- and with them the kernel's output -- are Pillow's bit for bit."""
    scale = float(in_size) / out_size
    filterscale = scale if scale >= 1.0 else 1.0
    support = 2.0 * filterscale
    ksize = int(np.ceil(support)) * 2 + 1
    ss = 1.0 / filterscale
    center = (np.arange(out_size, dtype=np.float64) + 0.5) * scale
    xmin = np.maximum(np.trunc(center - support + 0.5), 0.0).astype(np.int64)
    xmax = np.minimum(np.trunc(center + support + 0.5), float(in_size)).astype(np.int64) - xmin
    x = np.arange(ksize, dtype=np.float64)[None, :]
    t = np.abs((x + xmin[:, None] - center[:, None] + 0.5) * ss)
    a = -0.5
    w = np.where(t < 1.0, ((a + 2.0) * t - (a + 3.0)) * t * t + 1, np.where(t < 2.0, (((t - 5) * t + 8) * t - 4) * a, 0.0))
    w = np.where(np.arange(ksize)[None, :] < xmax[:, None], w, 0.0)
    ww = np.cumsum(w, axis=1)[:, -1:]                                   # sequential, like the C loop
    w = np.where(ww != 0.0, w / np.where(ww != 0.0, ww, 1.0), w)
    fixed = w * float(1 << 22)
    taps = np.where(w < 0, np.trunc(-0.5 + fixed), np.trunc(0.5 + fixed)).astype(np.int32)
    bounds = np.stack([xmin, xmax], axis=1).astype(np.int32)
    return bounds, taps, ksize


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
                 device: Union[str, torch.device] = "cuda", dtype: torch.dtype = torch.float32, native: Union[bool, None] = None):
        """native: None = the HIP kernel on a GPU (an error if libpvnative.so is missing there), PyTorch on the CPU;
        False = always the PyTorch path."""
        self.size, self.device, self.dtype = image_size, torch.device(device), dtype
        self.native = native
        self.mean = torch.tensor(mean, dtype=torch.float32, device=self.device)[None, :, None, None]
        self.std = torch.tensor(std, dtype=torch.float32, device=self.device)[None, :, None, None]
        self._mean3, self._std3 = tuple(float(v) for v in mean), tuple(float(v) for v in std)
        self._tables = {}                   # (H, W) -> device tap tables of the native kernel
        self.last_native = False

    def _native_ok(self) -> bool:
        from . import _native as N
        if self.native is False or self.device.type != "cuda" or self.dtype not in (torch.float32, torch.bfloat16):
            return False
        if not N.available():
            raise RuntimeError("GpuClipTransform on a GPU needs libpvnative.so (python -m vit_prisma_amd.build); "
                               "pass native=False for the PyTorch path")
        return True

    def _native_batch(self, x_u8: torch.Tensor) -> torch.Tensor:
        """``[B, H, W, 3]`` uint8 on the device -> ``[B, 3, S, S]`` through pv_clip_preprocess."""
        import ctypes as C
        from . import _native as N
        lib = N.lib()
        B, H, W, _ = x_u8.shape
        nw, nh = _resized_size(W, H, self.size)
        key = (H, W)
        if key not in self._tables:
            xb, xk, kx = _pil_coeffs(W, nw)
            yb, yk, ky = _pil_coeffs(H, nh)
            dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(self.device)   # noqa: E731
            self._tables[key] = (dev(xb), dev(xk), kx, dev(yb), dev(yk), ky)
        xb, xk, kx, yb, yk, ky = self._tables[key]
        left, top = int(round((nw - self.size) / 2.0)), int(round((nh - self.size) / 2.0))
        out = torch.empty(B, 3, self.size, self.size, dtype=self.dtype, device=self.device)
        x_u8 = x_u8.contiguous()
        m3, s3 = (C.c_float * 3)(*self._mean3), (C.c_float * 3)(*self._std3)
        N.check(lib.pv_clip_preprocess(x_u8.data_ptr(), B, H, W, xb.data_ptr(), xk.data_ptr(), kx, yb.data_ptr(), yk.data_ptr(), ky,
                                       nw, nh, left, top, self.size, m3, s3,
                                       N.PV_DTYPE_BF16 if self.dtype == torch.bfloat16 else N.PV_DTYPE_F32, out.data_ptr(),
                                       torch.cuda.current_stream(self.device).cuda_stream), "pv_clip_preprocess")
        self.last_native = True
        return out

    def one(self, img_u8_hwc: torch.Tensor) -> torch.Tensor:
        """``[H, W, 3]`` uint8 (host or device) -> ``[1, 3, S, S]`` normalised, on the device."""
        if img_u8_hwc.dtype == torch.uint8 and self._native_ok():
            return self._native_batch(img_u8_hwc.to(self.device, non_blocking=True)[None])
        self.last_native = False
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
            if images.dtype == torch.uint8 and self._native_ok():
                return self._native_batch(images.to(self.device, non_blocking=True))
            self.last_native = False
            x = images.to(self.device, non_blocking=True).permute(0, 3, 1, 2).float()
            h, w = x.shape[-2:]
            nw, nh = _resized_size(w, h, self.size)
            x = F.interpolate(x, size=(nh, nw), mode="bicubic", antialias=True, align_corners=False).round_().clamp_(0.0, 255.0)
            top, left = int(round((nh - self.size) / 2.0)), int(round((nw - self.size) / 2.0))
            x = x[:, :, top:top + self.size, left:left + self.size]
            return ((x / 255.0 - self.mean) / self.std).to(self.dtype)
        outs: List[torch.Tensor] = [self.one(im) for im in images]
        return torch.cat(outs, dim=0)
