"""TEST INFRASTRUCTURE ONLY (oracle/): numpy restatement of the reference's CLIP validation transform
(/root/reference/src/vit_prisma/transforms/model_transforms.py:9-20) for uint8 RGB images:

    Resize(image_size, BICUBIC, antialias) -> CenterCrop -> convert('RGB') -> ToTensor -> Normalize(mean, std)

The resize is not code of the reference tree: torchvision's ``Resize`` on a PIL image is ``Image.resize(size, BICUBIC)``,
i.e. Pillow's two-pass fixed-point resampler (third-party dependency, absent from /root/reference; this container has
Pillow 12.2.0, the algorithm below is unchanged since Pillow 3.x: ``src/libImaging/Resample.c`` -- ``precompute_coeffs``,
``normalize_coeffs_8bpc`` (PRECISION_BITS = 32 - 8 - 2), ``ImagingResampleHorizontal_8bpc`` /
``ImagingResampleVertical_8bpc``: horizontal pass first, each pass rounded and clipped to uint8).  Pinned by
tests/test_transform_oracle_cpu.py against Pillow itself (bit-exact on every tested size) -- Pillow IS what the reference runs.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import math
from typing import Sequence, Tuple

import numpy as np

PRECISION_BITS = 32 - 8 - 2
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def _bicubic(x: float) -> float:
    """Resample.c bicubic_filter, a = -0.5."""
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def precompute_coeffs(in_size: int, out_size: int) -> Tuple[np.ndarray, np.ndarray, int]:
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc for the whole axis [0, in_size) -> out_size samples.
    Returns (bounds [out, 2] int32 = (first input index, tap count), kk [out, ksize] int32 fixed-point taps, ksize)."""
    support_f = 2.0
    scale = float(in_size) / out_size
    filterscale = scale if scale >= 1.0 else 1.0
    support = support_f * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        k = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for w in k:
            ww += w
        for x in range(xmax):
            v = k[x] / ww if ww != 0.0 else k[x]
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk, ksize


def _resample_axis(img: np.ndarray, out_size: int, axis: int) -> np.ndarray:
    """One 8-bit pass along `axis` of an [H, W, C] uint8 image."""
    in_size = img.shape[axis]
    bounds, kk, _ = precompute_coeffs(in_size, out_size)
    src = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.empty((out_size,) + src.shape[1:], np.uint8)
    for xx in range(out_size):
        x0, n = int(bounds[xx, 0]), int(bounds[xx, 1])
        acc = np.tensordot(kk[xx, :n].astype(np.int64), src[x0:x0 + n], axes=(0, 0)) + (1 << (PRECISION_BITS - 1))
        out[xx] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def pil_resize_bicubic(img_hwc_u8: np.ndarray, new_w: int, new_h: int) -> np.ndarray:
    """Image.resize((new_w, new_h), BICUBIC) of an [H, W, C] uint8 image (ImagingResample: horizontal, then vertical)."""
    h, w = img_hwc_u8.shape[:2]
    out = img_hwc_u8
    if new_w != w:
        out = _resample_axis(out, new_w, 1)
    if new_h != h:
        out = _resample_axis(out, new_h, 0)
    return out


def resized_size(w: int, h: int, size: int) -> Tuple[int, int]:
    """torchvision Resize(int): the shorter side becomes `size`, the other int(size * long / short)."""
    if w <= h:
        return size, int(size * h / w)
    return int(size * w / h), size


def clip_val_transform(img_hwc_u8: np.ndarray, image_size: int = 224, mean: Sequence[float] = CLIP_MEAN,
                       std: Sequence[float] = CLIP_STD) -> np.ndarray:
    """[H, W, 3] uint8 -> [3, S, S] float32, the reference pipeline (model_transforms.py:14-20)."""
    h, w = img_hwc_u8.shape[:2]
    nw, nh = resized_size(w, h, image_size)
    r = pil_resize_bicubic(img_hwc_u8, nw, nh)
    left, top = int(round((nw - image_size) / 2.0)), int(round((nh - image_size) / 2.0))       # CenterCrop
    r = r[top:top + image_size, left:left + image_size]
    x = r.transpose(2, 0, 1).astype(np.float32) / np.float32(255.0)                            # ToTensor
    m = np.asarray(mean, np.float32)[:, None, None]
    s = np.asarray(std, np.float32)[:, None, None]
    return ((x - m) / s).astype(np.float32)                                                     # Normalize
