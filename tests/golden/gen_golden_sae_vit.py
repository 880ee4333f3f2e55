"""Golden fixture for HookedSAEViT (base_vit.py:827-1086) by EXECUTING THE REFERENCE (build container only):
a ReLU SAE and a top-k SAE spliced into the tiny model with the reference's own ``add_sae`` / ``reset_saes``.

    python tests/golden/gen_golden_sae_vit.py     ->  tests/golden/sae_vit_tiny.npz

Stored: the model output and the cache keys / selected cache tensors with one SAE attached (``hook_resid_post`` of block 0), with two
attached (+ ``hook_mlp_out`` of block 1), after ``reset_saes`` of one of them and after ``reset_saes()`` of all."""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from gen_golden_sae import ref_trainer_classes  # noqa: E402
from vit_prisma_amd.synth import ARCHS, synth_images, synth_sae_state, synth_vit_state  # noqa: E402

Cfg, SAE, _ = ref_trainer_classes()
from vit_prisma.configs.HookedViTConfig import HookedViTConfig  # noqa: E402
from vit_prisma.models.base_vit import HookedSAEViT  # noqa: E402

arch = ARCHS["tiny"]
model = HookedSAEViT(HookedViTConfig(**arch, dtype=torch.float32, device="cpu"))
model.load_state_dict({k: torch.from_numpy(v) for k, v in synth_vit_state(arch, 0).items()}, strict=True)
model.eval()
x = torch.from_numpy(synth_images(arch, 2, 1))


def make_sae(layer, subtype, act, kw, seed):
    cfg = Cfg(hook_point_layer=layer, layer_subtype=subtype, d_in=arch["d_model"], expansion_factor=4, activation_fn_str=act,
              activation_fn_kwargs=kw, normalize_activations="layer_norm", initialization_method="independent",
              b_dec_init_method="mean", _device="cpu", _dtype="float32", log_to_wandb=False, use_ghost_grads=False, verbose=False)
    sae = SAE(cfg)
    with torch.no_grad():
        for name, val in synth_sae_state(arch["d_model"], arch["d_model"] * 4, seed=seed).items():
            getattr(sae, name).copy_(torch.from_numpy(val))
    sae.eval()
    return sae


blob = {}


def snap(tag):
    with torch.no_grad():
        out, cache = model.run_with_cache(x)
    blob[f"{tag}_out"] = out.numpy()
    blob[f"{tag}_keys"] = np.array(list(cache.cache_dict.keys()))
    for k in ("blocks.0.hook_resid_post.hook_hidden_post", "blocks.0.hook_resid_post.hook_sae_out", "blocks.1.hook_resid_pre",
              "blocks.1.hook_mlp_out.hook_sae_in", "blocks.1.hook_resid_post"):
        if k in cache.cache_dict:
            blob[f"{tag}::{k}"] = cache.cache_dict[k].numpy()


snap("plain")
a = make_sae(0, "hook_resid_post", "relu", {}, 3)
b = make_sae(1, "hook_mlp_out", "topk", {"k": 8}, 4)
blob["hook_point_a"] = np.array(a.cfg.hook_point); blob["hook_point_b"] = np.array(b.cfg.hook_point)
model.add_sae(a)
snap("one")
with torch.no_grad():
    blob["one_forward"] = model(x).numpy()
model.add_sae(b)
snap("two")
model.reset_saes(a.cfg.hook_point)
snap("only_b")
model.reset_saes()
snap("reset")
np.savez_compressed(os.path.join(HERE, "sae_vit_tiny.npz"), **blob)
print({k: getattr(v, "shape", None) for k, v in blob.items()})

# ---- round 5: the same at CLIP ViT-B/32 size (bs = 4, fp32): a top-k SAE (768 -> 3072, k = 32) in place of blocks.6.hook_resid_post and a
# ReLU SAE in place of blocks.3.hook_mlp_out; every cache entry of the reference's run fingerprinted (oracle/vit_oracle.fingerprint)
#     -> tests/golden/sae_vit_b32_bs4.json
# (no bf16 budget: the reference's SAE config is fp32 only -- config.py:14-45 -- and a splice in another dtype than the model's takes the
# PyTorch path here)
import json
from oracle.vit_oracle import fingerprint  # noqa: E402

archB = ARCHS["clip-vit-b32"]
modelB = HookedSAEViT(HookedViTConfig(**archB, dtype=torch.float32, device="cpu"))
modelB.load_state_dict({k: torch.from_numpy(v) for k, v in synth_vit_state(archB, 0).items()}, strict=True)
modelB.eval()
xB = torch.from_numpy(synth_images(archB, 4, 1))


def make_sae_b(layer, subtype, act, kw, seed):
    cfg = Cfg(hook_point_layer=layer, layer_subtype=subtype, d_in=archB["d_model"], expansion_factor=4, activation_fn_str=act,
              activation_fn_kwargs=kw, normalize_activations="layer_norm", initialization_method="independent",
              b_dec_init_method="mean", _device="cpu", _dtype="float32", log_to_wandb=False, use_ghost_grads=False, verbose=False)
    sae = SAE(cfg)
    with torch.no_grad():
        for name, val in synth_sae_state(archB["d_model"], archB["d_model"] * 4, seed=seed).items():
            getattr(sae, name).copy_(torch.from_numpy(val))
    sae.eval()
    return sae


def snap_b():
    with torch.no_grad():
        out, cache = modelB.run_with_cache(xB)
    return {"keys": list(cache.cache_dict.keys()), "out": fingerprint(out.numpy()),
            "cache": {k: fingerprint(np.ascontiguousarray(v.numpy())) for k, v in cache.cache_dict.items()}}


big = {"arch": "clip-vit-b32", "batch": 4, "seed": 1,
       "saes": [{"layer": 6, "subtype": "hook_resid_post", "act": "topk", "kw": {"k": 32}, "seed": 7},
                {"layer": 3, "subtype": "hook_mlp_out", "act": "relu", "kw": {}, "seed": 8}]}
sa = make_sae_b(6, "hook_resid_post", "topk", {"k": 32}, 7)
sb = make_sae_b(3, "hook_mlp_out", "relu", {}, 8)
modelB.add_sae(sa)
big["one"] = snap_b()
modelB.add_sae(sb)
big["two"] = snap_b()
modelB.reset_saes()
with open(os.path.join(HERE, "sae_vit_b32_bs4.json"), "w") as f:
    json.dump(big, f)
print("sae_vit_b32_bs4.json", len(big["one"]["keys"]), len(big["two"]["keys"]), os.path.getsize(os.path.join(HERE, "sae_vit_b32_bs4.json")) // 1024, "kB")
