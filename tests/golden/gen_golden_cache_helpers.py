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
# the neuron / full decompositions (activation_cache.py:523-654, 737-826); get_neuron_results takes Slice objects only in the reference
from vit_prisma.prisma_tools.activation_cache import Slice  # noqa: E402
blob["neuron_results_l1"] = cache.get_neuron_results(1, Slice(None), Slice(None)).detach().numpy()
blob["neuron_results_l0_sliced"] = cache.get_neuron_results(0, neuron_slice=Slice((0, 8)), pos_slice=Slice(0)).detach().numpy()
st, labels = cache.stack_neuron_results(2, return_labels=True, incl_remainder=True)
blob["stack_neuron_results"] = st.detach().numpy(); blob["stack_neuron_results_labels"] = np.array(labels)
st, labels = cache.stack_neuron_results(2, apply_ln=True, pos_slice=0, neuron_slice=[1, 5, 7], return_labels=True)
blob["stack_neuron_results_ln_sliced"] = st.detach().numpy(); blob["stack_neuron_results_ln_sliced_labels"] = np.array(labels)
blob["stack_neuron_results_layer0"] = cache.stack_neuron_results(0).detach().numpy()
st, labels = cache.get_full_resid_decomposition(pos_slice=0, return_labels=True)
blob["full_resid_decomposition_pos0"] = st.detach().numpy(); blob["full_resid_decomposition_pos0_labels"] = np.array(labels)
st, labels = cache.get_full_resid_decomposition(layer=1, mlp_input=True, expand_neurons=False, apply_ln=True, pos_slice=3, return_labels=True)
blob["full_resid_decomposition_l1"] = st.detach().numpy(); blob["full_resid_decomposition_l1_labels"] = np.array(labels)
try:
    cache.get_full_resid_decomposition()
    blob["full_resid_decomposition_unsliced_raises"] = np.array(0)
except RuntimeError:
    blob["full_resid_decomposition_unsliced_raises"] = np.array(1)      # hook_embed has no CLS row: the cat cannot go through
np.savez_compressed(os.path.join(HERE, "cache_helpers_tiny.npz"), **blob)
print({k: v.shape for k, v in blob.items()})
