"""Real-weight loading for the two target architectures from LOCAL checkpoints (there is no network
on the build or GPU boxes, so nothing here downloads).

``convert_open_clip_weights`` / ``convert_hf_clip_weights`` produce the Prisma state-dict layout
(``cls_token [1,1,d]``, ``pos_embed.W_pos [T,d]``, ``embed.proj.weight [d,C,p,p]``, ``blocks.L.attn.W_Q [H,d,dh]``,
``W_O [H,dh,d]``, ``mlp.W_in [d,dmlp]`` ...) from an open_clip ``visual.*`` state dict
(/root/reference/src/vit_prisma/models/weight_conversion.py:276-313 + 345-429) or a HuggingFace ``CLIPModel``
``vision_model.*`` one (:147-273).  Same key names and tensor values as the reference's converters (pinned by
tests/test_weights_transforms_cpu.py against fixtures the reference's functions produced, and by a forward comparison with
an independent open_clip-style ViT written with torch's own multi-head attention).
"""
from __future__ import annotations

import os
from typing import Dict, Mapping, Optional

import torch

from .configs import HookedViTConfig


def _heads_in(w: torch.Tensor, cfg) -> torch.Tensor:
    """[(h dh), d] -> [h, d, dh]"""
    return w.reshape(cfg.n_heads, cfg.d_head, cfg.d_model).permute(0, 2, 1).contiguous()


def _heads_out(w: torch.Tensor, cfg) -> torch.Tensor:
    """[d, (h dh)] -> [h, dh, d]"""
    return w.reshape(cfg.d_model, cfg.n_heads, cfg.d_head).permute(1, 2, 0).contiguous()


def _block(sd: Mapping[str, torch.Tensor], cfg, new: str, q, k, v, bq, bk, bv, o, bo, ln1, ln2, fc1, fc2) -> Dict[str, torch.Tensor]:
    out = {
        f"{new}.ln1.w": sd[ln1 + ".weight"], f"{new}.ln1.b": sd[ln1 + ".bias"],
        f"{new}.ln2.w": sd[ln2 + ".weight"], f"{new}.ln2.b": sd[ln2 + ".bias"],
        f"{new}.attn.W_Q": _heads_in(q, cfg), f"{new}.attn.W_K": _heads_in(k, cfg), f"{new}.attn.W_V": _heads_in(v, cfg),
        f"{new}.attn.b_Q": bq.reshape(cfg.n_heads, cfg.d_head), f"{new}.attn.b_K": bk.reshape(cfg.n_heads, cfg.d_head),
        f"{new}.attn.b_V": bv.reshape(cfg.n_heads, cfg.d_head),
        f"{new}.attn.W_O": _heads_out(sd[o + ".weight"], cfg), f"{new}.attn.b_O": sd[o + ".bias"],
        f"{new}.mlp.W_in": sd[fc1 + ".weight"].t().contiguous(), f"{new}.mlp.b_in": sd[fc1 + ".bias"],
        f"{new}.mlp.W_out": sd[fc2 + ".weight"].t().contiguous(), f"{new}.mlp.b_out": sd[fc2 + ".bias"],
    }
    return out


def convert_open_clip_weights(old_state_dict: Mapping[str, torch.Tensor], cfg: HookedViTConfig) -> Dict[str, torch.Tensor]:
    sd = old_state_dict
    new: Dict[str, torch.Tensor] = {
        "cls_token": sd["visual.class_embedding"][None, None, :],
        "pos_embed.W_pos": sd["visual.positional_embedding"].clone(),
        "embed.proj.weight": sd["visual.conv1.weight"],
        "embed.proj.bias": torch.zeros(cfg.d_model),                     # open_clip's patch conv has no bias
        "ln_final.w": sd["visual.ln_post.weight"], "ln_final.b": sd["visual.ln_post.bias"],
        "ln_pre.w": sd["visual.ln_pre.weight"], "ln_pre.b": sd["visual.ln_pre.bias"],
        "head.W_H": sd["visual.proj"], "head.b_H": torch.zeros(cfg.n_classes),
    }
    for layer in range(cfg.n_layers):
        old = f"visual.transformer.resblocks.{layer}"
        q, k, v = sd[old + ".attn.in_proj_weight"].chunk(3)
        bq, bk, bv = sd[old + ".attn.in_proj_bias"].chunk(3)
        new.update(_block(sd, cfg, f"blocks.{layer}", q, k, v, bq, bk, bv, old + ".attn.out_proj", None, old + ".ln_1",
                          old + ".ln_2", old + ".mlp.c_fc", old + ".mlp.c_proj"))
    return new


def convert_hf_clip_weights(old_state_dict: Mapping[str, torch.Tensor], cfg: HookedViTConfig) -> Dict[str, torch.Tensor]:
    """HuggingFace ``CLIPModel`` (e.g. openai/clip-vit-large-patch14-336): ``vision_model.*`` + ``visual_projection``."""
    sd = old_state_dict
    new: Dict[str, torch.Tensor] = {
        "cls_token": sd["vision_model.embeddings.class_embedding"][None, None, :],
        "pos_embed.W_pos": sd["vision_model.embeddings.position_embedding.weight"],
        "embed.proj.weight": sd["vision_model.embeddings.patch_embedding.weight"],
        "embed.proj.bias": torch.zeros(cfg.d_model),
        "ln_final.w": sd["vision_model.post_layernorm.weight"], "ln_final.b": sd["vision_model.post_layernorm.bias"],
        "ln_pre.w": sd["vision_model.pre_layrnorm.weight"], "ln_pre.b": sd["vision_model.pre_layrnorm.bias"],    # (sic: HF's key)
        "head.W_H": sd["visual_projection.weight"].t().contiguous(), "head.b_H": torch.zeros(cfg.n_classes),
    }
    for layer in range(cfg.n_layers):
        old = f"vision_model.encoder.layers.{layer}"
        a = old + ".self_attn"
        new.update(_block(sd, cfg, f"blocks.{layer}", sd[a + ".q_proj.weight"], sd[a + ".k_proj.weight"], sd[a + ".v_proj.weight"],
                          sd[a + ".q_proj.bias"], sd[a + ".k_proj.bias"], sd[a + ".v_proj.bias"], a + ".out_proj", None,
                          old + ".layer_norm1", old + ".layer_norm2", old + ".mlp.fc1", old + ".mlp.fc2"))
    return new


def read_checkpoint(path: str) -> Dict[str, torch.Tensor]:
    """A local ``.safetensors`` / ``.pt`` / ``.bin`` file -> flat state dict (``state_dict`` / ``model`` wrappers unwrapped)."""
    if not os.path.isfile(path):
        raise FileNotFoundError(path)
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        return load_file(path, device="cpu")
    blob = torch.load(path, map_location="cpu", weights_only=True)
    for key in ("state_dict", "model"):
        if isinstance(blob, dict) and key in blob and isinstance(blob[key], dict):
            blob = blob[key]
    return {k[len("module."):] if k.startswith("module.") else k: v for k, v in blob.items()}


def load_clip_vision_weights(model, path: str, source: Optional[str] = None, fold_value_biases: bool = False):
    """Load a local open_clip or HuggingFace CLIP checkpoint into a ``HookedViT`` built for the matching architecture
    (``source``: "open_clip" | "hf"; default: detected from the key names).  fold_value_biases: as the reference's loader does by
    default (``HookedViT.fold_value_biases``).  Returns the model."""
    sd = read_checkpoint(path)
    if source is None:
        source = "open_clip" if "visual.conv1.weight" in sd else ("hf" if "vision_model.embeddings.class_embedding" in sd else None)
    if source == "open_clip":
        new = convert_open_clip_weights(sd, model.cfg)
    elif source == "hf":
        new = convert_hf_clip_weights(sd, model.cfg)
    else:
        raise ValueError("unrecognised checkpoint layout (expected open_clip 'visual.*' or HuggingFace 'vision_model.*' keys)")
    dtype = next(model.parameters()).dtype
    new = {k: v.to(dtype) for k, v in new.items()}
    if fold_value_biases:
        new = model.fold_value_biases(new)
    model.load_state_dict(new, strict=True)
    return model
