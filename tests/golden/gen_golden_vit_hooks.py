"""Fixtures for mutating forward hooks by EXECUTING THE REFERENCE (build container only):
CLIP ViT-B/32, bs = 4, the reference's ``run_with_cache(fwd_hooks=[...])`` (hooked_root_module.py:176-287) with
  A  blocks.6.hook_resid_post  <- t * 0.5 + 1.0                (a replacing hook: what SAE substitution does)
  B  blocks.3.hook_attn_out    <- 0                            (zero-ablation) together with
     blocks.9.hook_resid_mid   <- in-place edit of the CLS row (returns None)
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


CASES = {
    "A": [("blocks.6.hook_resid_post", scale_shift)],
    "B": [("blocks.3.hook_attn_out", zero), ("blocks.9.hook_resid_mid", edit_cls)],
}
KEYS = ["blocks.3.hook_attn_out", "blocks.3.hook_resid_mid", "blocks.6.hook_resid_post", "blocks.7.hook_resid_pre",
        "blocks.7.attn.hook_pattern", "blocks.9.hook_resid_mid", "blocks.9.mlp.hook_post", "blocks.11.hook_resid_post",
        "hook_ln_final"]

if __name__ == "__main__":
    arch = ARCHS["clip-vit-b32"]
    imgs = synth_images(arch, 4, seed=1)
    res = {"arch": "clip-vit-b32", "batch": 4, "seed": 1, "keys": KEYS, "cases": {}}
    runs = {}
    for dt in (torch.float32, torch.bfloat16):
        model, _ = build_reference_model("clip-vit-b32", dtype=dt)
        for name, hooks in CASES.items():
            with torch.no_grad():
                out, cache = model.run_with_cache(torch.from_numpy(imgs).to(dt), fwd_hooks=hooks, names_filter=KEYS)
            runs[(name, dt)] = (out, {k: v for k, v in cache.cache_dict.items()})
    for name in CASES:
        o32, c32 = runs[(name, torch.float32)]
        o16, c16 = runs[(name, torch.bfloat16)]
        res["cases"][name] = {
            "out": fingerprint(o32.numpy()),
            "cache": {k: fingerprint(c32[k].numpy()) for k in KEYS},
            "bf16_budget": {**{k: float((c32[k].double() - c16[k].double()).norm() / c32[k].double().norm().clamp_min(1e-30)) for k in KEYS},
                            "__out__": float((o32.double() - o16.double()).norm() / o32.double().norm())},
        }
        print(name, {k: round(v, 5) for k, v in res["cases"][name]["bf16_budget"].items()})
    with open(os.path.join(HERE, "vit_b32_hooks_bs4.json"), "w") as f:
        json.dump(res, f)
    print("vit_b32_hooks_bs4.json", os.path.getsize(os.path.join(HERE, "vit_b32_hooks_bs4.json")) // 1024, "kB")
