"""The reference's fold_value_biases (models/base_vit.py:498-532; applied by its load_hooked_model by default, model_loader.py:286, 352-358)
on the tiny model's synthetic state dict (build container only): the b_O / b_V it leaves and the output / cache of the folded model.

    python tests/golden/gen_golden_fold_value_biases.py     ->  tests/golden/vit_tiny_fold_value_biases.npz"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
from gen_golden_vit import build_reference_model  # noqa: E402
from vit_prisma_amd.synth import synth_images, synth_vit_state  # noqa: E402

model, arch = build_reference_model("tiny")
sd = {k: torch.from_numpy(v.copy()) for k, v in synth_vit_state(arch, 0).items()}
model.load_and_process_state_dict(sd, fold_ln=False, center_writing_weights=False, fold_value_biases=True)
blob = {}
for k, v in model.state_dict().items():
    if k.endswith("attn.b_O") or k.endswith("attn.b_V"):
        blob[f"param::{k}"] = v.numpy().copy()
x = torch.from_numpy(synth_images(arch, 2, 1))
with torch.no_grad():
    out, cache = model.run_with_cache(x)
blob["out"] = out.numpy()
for k in ("blocks.0.attn.hook_v", "blocks.1.attn.hook_z", "blocks.1.hook_attn_out", "blocks.1.hook_resid_post"):
    blob[f"cache::{k}"] = cache.cache_dict[k].numpy()
np.savez_compressed(os.path.join(HERE, "vit_tiny_fold_value_biases.npz"), **blob)
print(sorted(blob), os.path.getsize(os.path.join(HERE, "vit_tiny_fold_value_biases.npz")) // 1024, "kB")
