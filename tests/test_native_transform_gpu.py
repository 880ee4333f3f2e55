"""SURVEY.md 8f row 4 on the GPU: the CLIP validation transform as one HIP kernel (pv_clip_preprocess) against the oracle's
restatement of the reference pipeline (Pillow's resampler + crop + ToTensor + Normalize) -- bit for bit."""
import numpy as np
import pytest
import torch

from oracle import transform_oracle as TO
from vit_prisma_amd.transforms import GpuClipTransform

pytestmark = pytest.mark.gpu


def _img(h, w, seed):
    rng = np.random.default_rng(seed)
    img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    img[: h // 3] = (rng.integers(0, 2, (h // 3, w, 3)) * 255).astype(np.uint8)          # hard edges: the negative lobes must clip alike
    return img


@pytest.mark.parametrize("h,w,size", [(375, 500, 224), (500, 375, 224), (224, 224, 224), (97, 61, 224), (1080, 1920, 224),
                                      (2200, 3000, 224), (480, 640, 336), (336, 500, 336)])
def test_native_transform_is_the_reference_pipeline_bit_for_bit(h, w, size):
    imgs = np.stack([_img(h, w, 7 * h + w + i) for i in range(3)])
    t = GpuClipTransform(size, device="cuda", dtype=torch.float32)
    got = t(torch.from_numpy(imgs))
    assert t.last_native and got.shape == (3, 3, size, size) and got.dtype == torch.float32
    for i in range(3):
        ref = TO.clip_val_transform(imgs[i], size)
        assert np.array_equal(got[i].cpu().numpy(), ref), (i, float(np.abs(got[i].cpu().numpy() - ref).max()))
    one = t.one(torch.from_numpy(imgs[1]))
    assert torch.equal(one[0], got[1])
    # bf16 output = the fp32 result rounded to nearest even, what `.to(torch.bfloat16)` of the reference's tensor gives
    t16 = GpuClipTransform(size, device="cuda", dtype=torch.bfloat16)
    assert torch.equal(t16(torch.from_numpy(imgs)), got.to(torch.bfloat16))


def test_native_transform_feeds_the_native_vit():
    """uint8 batch -> pv_clip_preprocess -> run_with_cache, all on the device."""
    from vit_prisma_amd import HookedViT, HookedViTConfig
    from vit_prisma_amd.synth import ARCHS
    arch = ARCHS["tiny"]
    model = HookedViT(HookedViTConfig(**arch, device="cuda")).cuda().eval().use_native(True)
    t = GpuClipTransform(arch["image_size"], device="cuda", dtype=torch.float32)
    x = t(torch.from_numpy(np.stack([_img(90, 120, s) for s in range(4)])))
    with torch.no_grad():
        out, cache = model.run_with_cache(x)
    assert t.last_native and model.last_run_native and bool(torch.isfinite(out).all())
