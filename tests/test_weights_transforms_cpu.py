"""real-weight conversion and the CLIP input transform.

* ``convert_open_clip_weights`` / ``convert_hf_clip_weights`` against the REFERENCE's converters on a seeded synthetic
  checkpoint (build container only; skipped where /root/reference is absent), and -- everywhere -- against an independent
  open_clip-style vision transformer written with torch's own ``multi_head_attention_forward``: HookedViT with the
  converted weights must reproduce its image embedding at atol 1e-4, the tolerance of the reference's own upstream-parity
  test (/root/reference/tests/test_loading_CLIP-ViT-B-32-DataComp-XL-s13B-b90K.py:14-111, which needs the network).
* ``get_clip_val_transforms`` (PIL, the reference's pipeline) vs ``GpuClipTransform`` (batched, on device).
"""
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from vit_prisma_amd import HookedViT, HookedViTConfig
from vit_prisma_amd.synth import ARCHS
from vit_prisma_amd.transforms import GpuClipTransform, get_clip_val_transforms
from vit_prisma_amd.weights import convert_hf_clip_weights, convert_open_clip_weights, load_clip_vision_weights

from conftest import ROOT


def tiny_cfg():
    return HookedViTConfig(**ARCHS["tiny"], device="cpu")


def synth_open_clip_state(cfg, seed=0):
    """A state dict with open_clip's ``visual.*`` key names and shapes."""
    g = torch.Generator().manual_seed(seed)
    d, L, dm, p, C = cfg.d_model, cfg.n_layers, cfg.d_mlp, cfg.patch_size, cfg.n_channels
    T = (cfg.image_size // p) ** 2 + 1
    r = lambda *s, std=0.05: torch.randn(*s, generator=g) * std       # noqa: E731
    sd = {"visual.class_embedding": r(d, std=0.3), "visual.positional_embedding": r(T, d, std=0.3),
          "visual.conv1.weight": r(d, C, p, p, std=(C * p * p) ** -0.5), "visual.ln_pre.weight": 1 + r(d), "visual.ln_pre.bias": r(d),
          "visual.ln_post.weight": 1 + r(d), "visual.ln_post.bias": r(d), "visual.proj": r(d, cfg.n_classes, std=d ** -0.5)}
    for l in range(L):
        k = f"visual.transformer.resblocks.{l}"
        sd.update({k + ".ln_1.weight": 1 + r(d), k + ".ln_1.bias": r(d), k + ".ln_2.weight": 1 + r(d), k + ".ln_2.bias": r(d),
                   k + ".attn.in_proj_weight": r(3 * d, d, std=d ** -0.5), k + ".attn.in_proj_bias": r(3 * d),
                   k + ".attn.out_proj.weight": r(d, d, std=d ** -0.5), k + ".attn.out_proj.bias": r(d),
                   k + ".mlp.c_fc.weight": r(dm, d, std=d ** -0.5), k + ".mlp.c_fc.bias": r(dm),
                   k + ".mlp.c_proj.weight": r(d, dm, std=dm ** -0.5), k + ".mlp.c_proj.bias": r(d)})
    return sd


def open_clip_style_forward(sd, cfg, x):
    """open_clip's VisionTransformer.forward restated with torch primitives (pre-LN residual blocks around
    nn.MultiheadAttention, cls pooling, ln_post, projection) -- independent of HookedViT's per-head einsum layout."""
    d, H = cfg.d_model, cfg.n_heads
    t = F.conv2d(x, sd["visual.conv1.weight"], stride=cfg.patch_size).flatten(2).transpose(1, 2)
    t = torch.cat([sd["visual.class_embedding"].expand(t.shape[0], 1, d), t], dim=1) + sd["visual.positional_embedding"]
    t = F.layer_norm(t, (d,), sd["visual.ln_pre.weight"], sd["visual.ln_pre.bias"], 1e-5)
    for l in range(cfg.n_layers):
        k = f"visual.transformer.resblocks.{l}"
        h = F.layer_norm(t, (d,), sd[k + ".ln_1.weight"], sd[k + ".ln_1.bias"], 1e-5).transpose(0, 1)       # [T, B, d]
        a, _ = F.multi_head_attention_forward(h, h, h, d, H, sd[k + ".attn.in_proj_weight"], sd[k + ".attn.in_proj_bias"], None, None,
                                              False, 0.0, sd[k + ".attn.out_proj.weight"], sd[k + ".attn.out_proj.bias"],
                                              need_weights=False)
        t = t + a.transpose(0, 1)
        h = F.layer_norm(t, (d,), sd[k + ".ln_2.weight"], sd[k + ".ln_2.bias"], 1e-5)
        h = F.gelu(h @ sd[k + ".mlp.c_fc.weight"].t() + sd[k + ".mlp.c_fc.bias"])
        t = t + h @ sd[k + ".mlp.c_proj.weight"].t() + sd[k + ".mlp.c_proj.bias"]
    pooled = F.layer_norm(t[:, 0], (d,), sd["visual.ln_post.weight"], sd["visual.ln_post.bias"], 1e-5)
    return F.normalize(pooled @ sd["visual.proj"], dim=-1)


def to_hf_names(sd, cfg):
    """The same weights under HuggingFace CLIPModel's key names (q/k/v split, transposed projection)."""
    d = cfg.d_model
    hf = {"vision_model.embeddings.class_embedding": sd["visual.class_embedding"],
          "vision_model.embeddings.position_embedding.weight": sd["visual.positional_embedding"],
          "vision_model.embeddings.patch_embedding.weight": sd["visual.conv1.weight"],
          "vision_model.pre_layrnorm.weight": sd["visual.ln_pre.weight"], "vision_model.pre_layrnorm.bias": sd["visual.ln_pre.bias"],
          "vision_model.post_layernorm.weight": sd["visual.ln_post.weight"], "vision_model.post_layernorm.bias": sd["visual.ln_post.bias"],
          "visual_projection.weight": sd["visual.proj"].t().contiguous()}
    for l in range(cfg.n_layers):
        o, n = f"visual.transformer.resblocks.{l}", f"vision_model.encoder.layers.{l}"
        w, b = sd[o + ".attn.in_proj_weight"], sd[o + ".attn.in_proj_bias"]
        for i, nm in enumerate(("q_proj", "k_proj", "v_proj")):
            hf[f"{n}.self_attn.{nm}.weight"], hf[f"{n}.self_attn.{nm}.bias"] = w[i * d:(i + 1) * d], b[i * d:(i + 1) * d]
        hf[n + ".self_attn.out_proj.weight"], hf[n + ".self_attn.out_proj.bias"] = sd[o + ".attn.out_proj.weight"], sd[o + ".attn.out_proj.bias"]
        for a_, b_ in (("layer_norm1", "ln_1"), ("layer_norm2", "ln_2"), ("mlp.fc1", "mlp.c_fc"), ("mlp.fc2", "mlp.c_proj")):
            hf[f"{n}.{a_}.weight"], hf[f"{n}.{a_}.bias"] = sd[f"{o}.{b_}.weight"], sd[f"{o}.{b_}.bias"]
    return hf


def test_converted_weights_reproduce_an_independent_open_clip_forward(tmp_path):
    cfg = tiny_cfg()
    sd = synth_open_clip_state(cfg)
    x = torch.randn(3, cfg.n_channels, cfg.image_size, cfg.image_size, generator=torch.Generator().manual_seed(1))
    want = open_clip_style_forward(sd, cfg, x)
    new = convert_open_clip_weights(sd, cfg)
    model = HookedViT(cfg)
    assert set(new) == set(model.state_dict())                       # identical key set: strict load
    model.load_state_dict(new, strict=True)
    with torch.no_grad():
        got = model.eval()(x)
    assert float((got - want).abs().max()) < 1e-4
    # the HuggingFace layout of the same weights converts to the same tensors
    hf = convert_hf_clip_weights(to_hf_names(sd, cfg), cfg)
    for k in new:
        assert torch.equal(hf[k], new[k]), k
    # a checkpoint file on disk, both container formats
    from safetensors.torch import save_file
    save_file({k: v.contiguous() for k, v in sd.items()}, str(tmp_path / "open_clip.safetensors"))
    torch.save({"state_dict": {"module." + k: v for k, v in sd.items()}}, str(tmp_path / "open_clip.pt"))
    for f in ("open_clip.safetensors", "open_clip.pt"):
        m2 = load_clip_vision_weights(HookedViT(cfg), str(tmp_path / f))
        with torch.no_grad():
            assert torch.equal(m2.eval()(x), got)


@pytest.mark.skipif(not os.path.isdir("/root/reference/src"), reason="the reference tree only exists in the build container")
def test_converters_equal_the_references_on_the_same_checkpoint():
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from _refimport import install
    install()
    from vit_prisma.models.weight_conversion import convert_open_clip_weights as ref_oc
    from vit_prisma.models.weight_conversion import convert_kandinsky_clip_weights as ref_hf      # (its HF CLIPModel converter)
    cfg = tiny_cfg()
    sd = synth_open_clip_state(cfg, seed=3)
    ours, ref = convert_open_clip_weights(sd, cfg), ref_oc(sd, cfg)
    assert set(ours) == set(ref)
    for k in ref:
        assert ours[k].shape == ref[k].shape and torch.equal(ours[k], ref[k]), k
    hf_sd = to_hf_names(sd, cfg)
    ours, ref = convert_hf_clip_weights(hf_sd, cfg), ref_hf(hf_sd, cfg, device="cpu")
    for k in ref:
        assert torch.equal(ours[k], ref[k]), k


def _smooth_image(h, w, seed=0):
    rs = np.random.RandomState(seed)
    y, x = np.mgrid[0:h, 0:w]
    im = np.stack([127 + 100 * np.sin(x / 23.0 + y / 31.0), 127 + 100 * np.cos(x / 17.0 - y / 41.0), 127 + 80 * np.sin((x + y) / 13.0)], -1)
    return np.clip(im + rs.randn(h, w, 3) * 6, 0, 255).astype(np.uint8)


def test_clip_val_transform_pil_and_device_pipelines_agree():
    from PIL import Image
    ref_t = get_clip_val_transforms()
    dev_t = GpuClipTransform(device="cpu")
    imgs = [_smooth_image(h, w, s) for s, (h, w) in enumerate([(300, 400), (512, 384), (224, 224), (700, 500)])]
    ref = torch.stack([ref_t(Image.fromarray(a)) for a in imgs])
    got = dev_t([torch.from_numpy(a.copy()) for a in imgs])
    assert got.shape == ref.shape == (4, 3, 224, 224)
    level = 1.0 / 255.0 / 0.26                                        # one uint8 level in normalised units
    assert float((got - ref).abs().max()) <= 1.5 * level               # PIL rounds to uint8 between its two passes
    assert torch.equal(got[2], ref[2])                                # already 224 x 224: no resampling at all
    # statistics of the reference pipeline: mean / std as documented
    one = ref_t(Image.fromarray(np.full((256, 256, 3), 128, np.uint8)))
    assert abs(float(one[0].mean()) - (128 / 255 - 0.48145466) / 0.26862954) < 1e-5
    # batched same-size path
    batch = torch.from_numpy(np.stack([_smooth_image(320, 480, 9), _smooth_image(320, 480, 10)]))
    assert torch.allclose(dev_t(batch), dev_t([batch[0], batch[1]]), atol=1e-6)
