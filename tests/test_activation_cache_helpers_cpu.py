"""ActivationCache analysis helpers against golden outputs produced by the reference's own ActivationCache
(tests/golden/gen_golden_cache_helpers.py) on the same tiny model / input."""
import os

import numpy as np
import torch

from conftest import GOLDEN, rel_fro
from test_vit_host_vs_golden import build
from vit_prisma_amd.synth import synth_images


def test_cache_helpers_match_reference():
    g = np.load(os.path.join(GOLDEN, "cache_helpers_tiny.npz"))
    model, arch = build("tiny")
    with torch.no_grad():
        _, cache = model.run_with_cache(torch.from_numpy(synth_images(arch, 2, 1)))
        acc, labels = cache.accumulated_resid(return_labels=True, incl_mid=True)
        assert labels == [str(s) for s in g["accumulated_resid_labels"]]
        assert rel_fro(acc.numpy(), g["accumulated_resid"]) < 1e-5
        assert rel_fro(cache.accumulated_resid(layer=1, apply_ln=True, mlp_input=True).numpy(), g["accumulated_resid_ln"]) < 1e-5
        dec, labels = cache.decompose_resid(return_labels=True, incl_embeds=False)
        assert labels == [str(s) for s in g["decompose_resid_labels"]]
        assert rel_fro(dec.numpy(), g["decompose_resid"]) < 1e-5
        got = cache.decompose_resid(layer=2, mode="attn", apply_ln=True, incl_embeds=False, pos_slice=0).numpy()
        assert got.shape == g["decompose_resid_attn_ln"].shape and rel_fro(got, g["decompose_resid_attn_ln"]) < 1e-5
        heads, labels = cache.stack_head_results(return_labels=True, incl_remainder=True)
        assert labels == [str(s) for s in g["stack_head_results_labels"]]
        assert heads.shape == g["stack_head_results"].shape and rel_fro(heads.numpy(), g["stack_head_results"]) < 1e-5
        assert rel_fro(cache.stack_activation("pattern").numpy(), g["stack_activation_pattern"]) < 1e-5
        assert rel_fro(cache.apply_ln_to_stack(cache.accumulated_resid(layer=1), layer=1).numpy(), g["apply_ln_to_stack"]) < 1e-5


def test_neuron_and_full_decompositions_match_reference():
    """get_neuron_results / stack_neuron_results / get_full_resid_decomposition (activation_cache.py:523-654, 737-826) against
    what the reference's own ActivationCache returned on the same tiny model and input -- corner cases included (no layers below,
    a collapsing position slice, and the un-sliced full decomposition, which cannot be concatenated for a ViT with a CLS token:
    hook_embed holds the patch rows only)."""
    import pytest
    from vit_prisma_amd.utils import Slice
    g = np.load(os.path.join(GOLDEN, "cache_helpers_tiny.npz"))
    model, arch = build("tiny")
    with torch.no_grad():
        _, cache = model.run_with_cache(torch.from_numpy(synth_images(arch, 2, 1)))
        assert rel_fro(cache.get_neuron_results(1).numpy(), g["neuron_results_l1"]) < 1e-5
        assert rel_fro(cache.get_neuron_results(1, Slice(None), Slice(None)).numpy(), g["neuron_results_l1"]) < 1e-5
        got = cache.get_neuron_results(0, neuron_slice=(0, 8), pos_slice=0).numpy()
        assert got.shape == g["neuron_results_l0_sliced"].shape and rel_fro(got, g["neuron_results_l0_sliced"]) < 1e-5
        st, labels = cache.stack_neuron_results(2, return_labels=True, incl_remainder=True)
        assert labels == [str(s) for s in g["stack_neuron_results_labels"]]
        assert st.shape == g["stack_neuron_results"].shape and rel_fro(st.numpy(), g["stack_neuron_results"]) < 1e-5
        # the components add up to the stream they decompose
        assert rel_fro(st.sum(0).numpy(), cache[("resid_post", 1)].numpy()) < 1e-5
        st, labels = cache.stack_neuron_results(2, apply_ln=True, pos_slice=0, neuron_slice=[1, 5, 7], return_labels=True)
        assert labels == [str(s) for s in g["stack_neuron_results_ln_sliced_labels"]]
        assert st.shape == g["stack_neuron_results_ln_sliced"].shape and rel_fro(st.numpy(), g["stack_neuron_results_ln_sliced"]) < 1e-5
        assert tuple(cache.stack_neuron_results(0).shape) == tuple(g["stack_neuron_results_layer0"].shape)
        rem = cache.stack_neuron_results(0, incl_remainder=True)
        assert isinstance(rem, list) and len(rem) == 1                     # (the reference's corner case, kept)
        st, labels = cache.get_full_resid_decomposition(pos_slice=0, return_labels=True)
        assert labels == [str(s) for s in g["full_resid_decomposition_pos0_labels"]]
        assert st.shape == g["full_resid_decomposition_pos0"].shape and rel_fro(st.numpy(), g["full_resid_decomposition_pos0"]) < 1e-5
        st, labels = cache.get_full_resid_decomposition(layer=1, mlp_input=True, expand_neurons=False, apply_ln=True, pos_slice=3,
                                                        return_labels=True)
        assert labels == [str(s) for s in g["full_resid_decomposition_l1_labels"]]
        assert st.shape == g["full_resid_decomposition_l1"].shape and rel_fro(st.numpy(), g["full_resid_decomposition_l1"]) < 1e-5
        assert int(g["full_resid_decomposition_unsliced_raises"]) == 1
        with pytest.raises(RuntimeError):
            cache.get_full_resid_decomposition()
