"""Golden outputs of the reference's ActivationCache analysis helpers on the tiny model (build container
only): executes /root/reference's ActivationCache on a cache produced by the reference HookedViT."""
import os, sys
import numpy as np, torch
HERE = os.path.dirname(os.path.abspath(__file__)); ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, HERE)
from gen_golden_vit import build_reference_model, run_ref
from vit_prisma_amd.synth import synth_images

model, arch = build_reference_model("tiny")
out, cache = run_ref(model, synth_images(arch, 2, 1))
blob = {}
acc, labels = cache.accumulated_resid(return_labels=True, incl_mid=True)
blob["accumulated_resid"] = acc.numpy(); blob["accumulated_resid_labels"] = np.array(labels)
blob["accumulated_resid_ln"] = cache.accumulated_resid(layer=1, apply_ln=True, mlp_input=True).numpy()
dec, labels = cache.decompose_resid(return_labels=True, incl_embeds=False)
blob["decompose_resid"] = dec.numpy(); blob["decompose_resid_labels"] = np.array(labels)
blob["decompose_resid_attn_ln"] = cache.decompose_resid(layer=2, mode="attn", apply_ln=True, incl_embeds=False, pos_slice=0).numpy()
heads, labels = cache.stack_head_results(return_labels=True, incl_remainder=True)
blob["stack_head_results"] = heads.detach().numpy(); blob["stack_head_results_labels"] = np.array(labels)
blob["stack_activation_pattern"] = cache.stack_activation("pattern").numpy()
blob["apply_ln_to_stack"] = cache.apply_ln_to_stack(cache.accumulated_resid(layer=1), layer=1, pos_slice=(0, 5)).numpy() if False else cache.apply_ln_to_stack(cache.accumulated_resid(layer=1), layer=1).numpy()
np.savez_compressed(os.path.join(HERE, "cache_helpers_tiny.npz"), **blob)
print({k: v.shape for k, v in blob.items()})
