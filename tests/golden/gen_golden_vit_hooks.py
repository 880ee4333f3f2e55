"""Fixtures for mutating forward hooks by EXECUTING THE REFERENCE (build container only):
CLIP ViT-B/32, bs = 4, the reference's ``run_with_cache(fwd_hooks=[...])`` (hooked_root_module.py:176-287) with
  A  blocks.6.hook_resid_post  <- t * 0.5 + 1.0                (a replacing hook: what SAE substitution does)
  B  blocks.3.hook_attn_out    <- 0                            (zero-ablation) together with
     blocks.9.hook_resid_mid   <- in-place edit of the CLS row (returns None)
and, INSIDE a block (the split positions of pv_vit_forward_stage; hook_point.py:44-45, attention.py:135-152, 267-281):
  C  blocks.5.attn.hook_z           <- head 3 zeroed in place   (head ablation)
  D  blocks.4.attn.hook_pattern     <- nobody attends to the CLS token, rows renormalised
  E  blocks.7.attn.hook_attn_scores <- head 0 cannot see the last key (-inf, in place)
  F  blocks.8.mlp.hook_post         <- every third neuron zeroed (neuron ablation)
  G  blocks.2.ln1.hook_scale        <- 2.0 everywhere           ("frozen LayerNorm")
each with its own key list (the hooked tensor where it is finite, what the block computes behind it, the stream after it).
Writes vit_b32_hooks_bs4.json: fingerprints (oracle.vit_oracle.fingerprint) of the output and of selected cache tensors in
fp32, and the reference's own bf16-vs-fp32 error (rel-Frobenius) for the same keys -- the budget the bf16 HIP path is held
to under hooks.
    python tests/golden/gen_golden_vit_hooks.py
"""
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
from gen_golden_vit import build_reference_model  # noqa: E402
from oracle.vit_oracle import fingerprint  # noqa: E402
from vit_prisma_amd.synth import ARCHS, synth_images  # noqa: E402


def scale_shift(t, hook):
    return t * 0.5 + 1.0


def zero(t, hook):
    return torch.zeros_like(t)


def edit_cls(t, hook):
    t[:, 0] = 0.25


def kill_head_3(t, hook):                 # [B, T, H, dh], in place
    t[:, :, 3] = 0.0


def no_cls_attention(t, hook):            # pattern [B, H, T, T]
    t = t.clone()
    t[..., 0] = 0.0
    return t / t.sum(-1, keepdim=True).clamp_min(1e-6)


def mask_last_key(t, hook):               # scores [B, H, T, T], in place
    t[:, 0, :, -1] = float("-inf")


def kill_neurons(t, hook):                # [B, T, d_mlp], in place
    t[..., ::3] = 0.0


def freeze_scale(t, hook):                # [B, T, 1]
    return torch.full_like(t, 2.0)


CASES = {
    "A": [("blocks.6.hook_resid_post", scale_shift)],
    "B": [("blocks.3.hook_attn_out", zero), ("blocks.9.hook_resid_mid", edit_cls)],
    "C": [("blocks.5.attn.hook_z", kill_head_3)],
    "D": [("blocks.4.attn.hook_pattern", no_cls_attention)],
    "E": [("blocks.7.attn.hook_attn_scores", mask_last_key)],
    "F": [("blocks.8.mlp.hook_post", kill_neurons)],
    "G": [("blocks.2.ln1.hook_scale", freeze_scale)],
}
KEYS = ["blocks.3.hook_attn_out", "blocks.3.hook_resid_mid", "blocks.6.hook_resid_post", "blocks.7.hook_resid_pre",
        "blocks.7.attn.hook_pattern", "blocks.9.hook_resid_mid", "blocks.9.mlp.hook_post", "blocks.11.hook_resid_post",
        "hook_ln_final"]
CASE_KEYS = {
    "C": ["blocks.5.attn.hook_pattern", "blocks.5.attn.hook_z", "blocks.5.hook_attn_out", "blocks.5.hook_resid_mid",
          "blocks.5.hook_resid_post", "blocks.11.hook_resid_post", "hook_ln_final"],
    "D": ["blocks.4.attn.hook_attn_scores", "blocks.4.attn.hook_pattern", "blocks.4.attn.hook_z", "blocks.4.hook_attn_out",
          "blocks.4.hook_resid_post", "blocks.11.hook_resid_post", "hook_ln_final"],
    "E": ["blocks.7.attn.hook_pattern", "blocks.7.attn.hook_z", "blocks.7.hook_attn_out", "blocks.7.hook_resid_post",
          "blocks.11.hook_resid_post", "hook_ln_final"],
    "F": ["blocks.8.mlp.hook_pre", "blocks.8.mlp.hook_post", "blocks.8.hook_mlp_out", "blocks.8.hook_resid_post",
          "blocks.11.hook_resid_post", "hook_ln_final"],
    "G": ["blocks.2.ln1.hook_scale", "blocks.2.ln1.hook_normalized", "blocks.2.attn.hook_q", "blocks.2.attn.hook_pattern",
          "blocks.2.hook_attn_out", "blocks.2.hook_resid_post", "blocks.11.hook_resid_post", "hook_ln_final"],
}

if __name__ == "__main__":
    arch = ARCHS["clip-vit-b32"]
    imgs = synth_images(arch, 4, seed=1)
    res = {"arch": "clip-vit-b32", "batch": 4, "seed": 1, "keys": KEYS, "cases": {}}
    runs = {}
    for dt in (torch.float32, torch.bfloat16):
        model, _ = build_reference_model("clip-vit-b32", dtype=dt)
        for name, hooks in CASES.items():
            with torch.no_grad():
                out, cache = model.run_with_cache(torch.from_numpy(imgs).to(dt), fwd_hooks=hooks, names_filter=CASE_KEYS.get(name, KEYS))
            runs[(name, dt)] = (out, {k: v for k, v in cache.cache_dict.items()})
    for name in CASES:
        o32, c32 = runs[(name, torch.float32)]
        o16, c16 = runs[(name, torch.bfloat16)]
        keys = CASE_KEYS.get(name, KEYS)
        assert list(c32.keys()) == keys
        res["cases"][name] = {
            "keys": keys,
            "out": fingerprint(o32.numpy()),
            "cache": {k: fingerprint(c32[k].numpy()) for k in keys},
            "bf16_budget": {**{k: float((c32[k].double() - c16[k].double()).norm() / c32[k].double().norm().clamp_min(1e-30)) for k in keys},
                            "__out__": float((o32.double() - o16.double()).norm() / o32.double().norm())},
        }
        print(name, {k: round(v, 5) for k, v in res["cases"][name]["bf16_budget"].items()})
    with open(os.path.join(HERE, "vit_b32_hooks_bs4.json"), "w") as f:
        json.dump(res, f)
    print("vit_b32_hooks_bs4.json", os.path.getsize(os.path.join(HERE, "vit_b32_hooks_bs4.json")) // 1024, "kB")
