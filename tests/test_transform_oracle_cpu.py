"""The transform oracle (oracle/transform_oracle.py) is Pillow's resampler restated; Pillow is what the reference's
``get_clip_val_transforms`` runs (torchvision ``Resize`` on a PIL image), so the oracle is pinned to Pillow itself, bit for bit,
and to the package's PIL pipeline (``vit_prisma_amd.transforms.get_clip_val_transforms``) end to end."""
import numpy as np
import pytest
import torch

from oracle import transform_oracle as TO

SIZES = [(375, 500), (500, 375), (224, 224), (256, 341), (1080, 1920), (97, 61), (640, 480), (336, 336), (200, 4000 // 13)]


@pytest.mark.parametrize("h,w", SIZES)
@pytest.mark.parametrize("size", [224, 336])
def test_resize_is_pillows_bit_for_bit(h, w, size):
    from PIL import Image
    rng = np.random.default_rng(h * 10007 + w + size)
    img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    img[: h // 3] = (rng.integers(0, 2, (h // 3, w, 3)) * 255).astype(np.uint8)          # hard edges: ringing must clip alike
    nw, nh = TO.resized_size(w, h, size)
    ref = np.asarray(Image.fromarray(img).resize((nw, nh), Image.BICUBIC))
    got = TO.pil_resize_bicubic(img, nw, nh)
    assert got.shape == ref.shape and np.array_equal(got, ref)


@pytest.mark.parametrize("h,w", [(375, 500), (500, 375), (97, 61)])
def test_full_transform_equals_the_pil_pipeline(h, w):
    from PIL import Image
    from vit_prisma_amd.transforms import get_clip_val_transforms
    img = np.random.default_rng(h + w).integers(0, 256, (h, w, 3), dtype=np.uint8)
    ref = get_clip_val_transforms(224)(Image.fromarray(img)).numpy()
    got = TO.clip_val_transform(img, 224)
    assert got.dtype == np.float32 and np.array_equal(got, ref)


@pytest.mark.parametrize("n_in,n_out", [(500, 298), (375, 224), (224, 224), (61, 224), (1920, 398), (4000, 336), (97, 356), (3, 224)])
def test_the_packages_tap_tables_are_the_oracles(n_in, n_out):
    """vit_prisma_amd.transforms._pil_coeffs (vectorised, feeds the HIP kernel) == the oracle's scalar restatement of
    Resample.c, bit for bit."""
    from vit_prisma_amd.transforms import _pil_coeffs
    b0, k0, ks0 = TO.precompute_coeffs(n_in, n_out)
    b1, k1, ks1 = _pil_coeffs(n_in, n_out)
    assert ks0 == ks1 and np.array_equal(b0, b1) and np.array_equal(k0, k1)
