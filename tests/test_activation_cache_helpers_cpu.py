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
