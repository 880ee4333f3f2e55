"""Golden caches of the reference run with its flag-gated HookPoints enabled (build container only): use_attn_result,
use_split_qkv_input, use_attn_in, use_hook_mlp_in (transformer_block.py:88-129, attention.py:155-183) on the tiny model, fp32.

    python tests/golden/gen_golden_vit_flags.py     ->  tests/golden/vit_tiny_flags.npz

Two configurations: "all" (the four flags) and "result_mlp" (use_attn_result + use_hook_mlp_in: no head dimension on the block
inputs).  Every cache tensor, the key order and the output."""
import os, sys
import numpy as np, torch
HERE = os.path.dirname(os.path.abspath(__file__)); ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, HERE)
from gen_golden_vit import build_reference_model, run_ref
from vit_prisma_amd.synth import synth_images

blob = {}
for tag, flags in (("all", dict(use_attn_result=True, use_split_qkv_input=True, use_attn_in=True, use_hook_mlp_in=True)),
                   ("result_mlp", dict(use_attn_result=True, use_hook_mlp_in=True)),
                   ("attn_in", dict(use_attn_in=True)), ("split", dict(use_split_qkv_input=True))):
    model, arch = build_reference_model("tiny")
    for k, v in flags.items():
        setattr(model.cfg, k, v)
    out, cache = run_ref(model, synth_images(arch, 2, 1))
    blob[f"{tag}::__out__"] = out.numpy()
    blob[f"{tag}::__keys__"] = np.array(list(cache.cache_dict.keys()))
    for k, v in cache.cache_dict.items():
        if tag in ("all", "result_mlp") or k.startswith("blocks.0.hook_") or k.startswith("blocks.0.ln1."):    # (the others: keys only)
            blob[f"{tag}::{k}"] = np.ascontiguousarray(v.numpy())
    print(tag, len(cache.cache_dict), [(k, tuple(v.shape)) for k, v in cache.cache_dict.items() if k.startswith("blocks.0.")])
np.savez_compressed(os.path.join(HERE, "vit_tiny_flags.npz"), **blob)
print(os.path.getsize(os.path.join(HERE, "vit_tiny_flags.npz")) // 1024, "kB")

# ---- the same at CLIP ViT-B/32 size (bs = 2, fp32, the four flags): fingerprints only (oracle/vit_oracle.fingerprint) -> vit_b32_flags_bs2.json
import json
from oracle.vit_oracle import fingerprint
model, arch = build_reference_model("clip-vit-b32")
for k in ("use_attn_result", "use_split_qkv_input", "use_attn_in", "use_hook_mlp_in"):
    setattr(model.cfg, k, True)
out, cache = run_ref(model, synth_images(arch, 2, 1))
big = {"keys": list(cache.cache_dict.keys()), "out": fingerprint(out.numpy()),
       "cache": {k: fingerprint(v.numpy()) for k, v in cache.cache_dict.items()}}
with open(os.path.join(HERE, "vit_b32_flags_bs2.json"), "w") as f:
    json.dump(big, f)
print("b32 flags", len(big["keys"]), os.path.getsize(os.path.join(HERE, "vit_b32_flags_bs2.json")) // 1024, "kB")

# ---- round 5: the reference's OWN bf16 run with the four flags against its fp32 run (same images): the per-key error budget the bf16 HIP
# mode is held to on the flag-gated entries too (as vit_b32_bf16_budget*.json does for the 214 plain entries) -> vit_b32_flags_bf16_budget_bs2.json
model16, _ = build_reference_model("clip-vit-b32", dtype=torch.bfloat16)
for k in ("use_attn_result", "use_split_qkv_input", "use_attn_in", "use_hook_mlp_in"):
    setattr(model16.cfg, k, True)
out16, cache16 = run_ref(model16, synth_images(arch, 2, 1))
assert list(cache16.cache_dict.keys()) == big["keys"]
budget = {}
for k, v32 in cache.cache_dict.items():
    a, b_ = v32.double(), cache16.cache_dict[k].double()
    budget[k] = {"rel_fro": float((a - b_).norm() / a.norm().clamp_min(1e-30)), "dtype_bf16_run": str(cache16.cache_dict[k].dtype)}
budget["__out__"] = {"rel_fro": float((out.double() - out16.double()).norm() / out.double().norm()), "dtype_bf16_run": str(out16.dtype)}
with open(os.path.join(HERE, "vit_b32_flags_bf16_budget_bs2.json"), "w") as f:
    json.dump({"arch": "clip-vit-b32", "batch": 2, "seed": 1, "flags": ["use_attn_result", "use_split_qkv_input", "use_attn_in", "use_hook_mlp_in"],
               "budget": budget}, f)
rel = sorted(v["rel_fro"] for v in budget.values())
print("b32 flags bf16 budget rel_fro min / median / max", rel[0], rel[len(rel) // 2], rel[-1], os.path.getsize(os.path.join(HERE, "vit_b32_flags_bf16_budget_bs2.json")) // 1024, "kB")
