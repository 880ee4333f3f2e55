"""Golden runs of the reference's HookedSAEViT (base_vit.py:827-1086) with an SAE spliced where the forward STARTS (build container only):
in place of ``blocks.0.hook_resid_pre`` (the first block's entry) -- tiny model, fp32 -- and the same SAE after ``cfg.hook_point = "hook_embed"``:
the reference's setter stores a value its getter never reads (sae/config.py:428-436), so the second case IS the first (same keys, same
numbers): an SAE cannot be moved to the embedding stage that way, and the drop-in must not move it either.

    python tests/golden/gen_golden_sae_vit_edges.py     ->  tests/golden/sae_vit_tiny_edges.npz

Per case: the output, the key order and every cache tensor of run_with_cache."""
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
x = torch.from_numpy(synth_images(arch, 2, 1))
blob = {}
for tag, hook_point in (("entry0", "blocks.0.hook_resid_pre"), ("embed", "hook_embed")):
    model = HookedSAEViT(HookedViTConfig(**arch, dtype=torch.float32, device="cpu"))
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth_vit_state(arch, 0).items()}, strict=True)
    model.eval()
    cfg = Cfg(hook_point_layer=0, layer_subtype="hook_resid_pre", d_in=arch["d_model"], expansion_factor=4, activation_fn_str="relu",
              activation_fn_kwargs={}, normalize_activations="layer_norm", initialization_method="independent", b_dec_init_method="mean",
              _device="cpu", _dtype="float32", log_to_wandb=False, use_ghost_grads=False, verbose=False)
    sae = SAE(cfg)
    with torch.no_grad():
        for name, val in synth_sae_state(arch["d_model"], arch["d_model"] * 4, seed=5).items():
            getattr(sae, name).copy_(torch.from_numpy(val))
    sae.eval()
    sae.cfg.hook_point = hook_point
    model.add_sae(sae)
    with torch.no_grad():
        out, cache = model.run_with_cache(x)
    blob[f"{tag}::__out__"] = out.numpy()
    blob[f"{tag}::__keys__"] = np.array(list(cache.cache_dict.keys()))
    for k, v in cache.cache_dict.items():
        blob[f"{tag}::{k}"] = np.ascontiguousarray(v.numpy())
    print(tag, hook_point, len(cache.cache_dict), float(out.abs().sum()))
np.savez_compressed(os.path.join(HERE, "sae_vit_tiny_edges.npz"), **blob)
print(os.path.getsize(os.path.join(HERE, "sae_vit_tiny_edges.npz")) // 1024, "kB")
