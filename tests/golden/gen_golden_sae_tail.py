"""Golden fixtures for the tail of SURVEY.md 8(f) row 3 (round 6), by EXECUTING THE REFERENCE (build container only): its
StandardSparseAutoencoder through its own VisionSAETrainer.train_step (/root/reference/src/vit_prisma/sae/train_sae.py:278-411) for 3 steps
at d_in = 64, d_sae = 512, N = 256 with

    relu_constnorm   normalize_activations = "constant_norm_rescale" (sae/sae.py:60-72), ReLU + L1
    topk_constnorm   the same normalisation on the top-k SAE (k = 8)
    tanh_relu        activation_fn_str = "tanh-relu" (sae/sae.py:823-830), L1 term
    relu_lp2         lp_norm = 2 in the sparsity term (sae/sae.py:617)
    tc_topk_ghost    a Transcoder (skip connection) with use_ghost_grads, top-k (sae/transcoder.py:66-116 + sae/sae.py:151-179)
    tc_relu_ghost    the same on the ReLU + L1 form, without the skip connection

    python tests/golden/gen_golden_sae_tail.py     ->  tests/golden/sae_tail_steps.npz

Same layout as sae_variants_steps.npz (gen_golden_sae_variants.run): per variant and step the loss / mse / l1 / l0 scalars, act_freq,
n_since_fired; parameters after step 3; the initial parameters.
"""
from __future__ import annotations

import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

from gen_golden_sae_variants import run  # noqa: E402

VARIANTS = {
    "relu_constnorm": dict(activation_fn_str="relu", activation_fn_kwargs={}, l1_coefficient=2e-3, normalize_activations="constant_norm_rescale"),
    "topk_constnorm": dict(activation_fn_str="topk", activation_fn_kwargs={"k": 8}, normalize_activations="constant_norm_rescale"),
    "tanh_relu": dict(activation_fn_str="tanh-relu", activation_fn_kwargs={}, l1_coefficient=2e-3),
    "relu_lp2": dict(activation_fn_str="relu", activation_fn_kwargs={}, l1_coefficient=2e-3, lp_norm=2),
    # ghost gradients on a Transcoder (transcoder.py:82-86 hands the ghost term the INPUT activation): top-k and ReLU + L1 forms
    "tc_topk_ghost": dict(is_transcoder=True, transcoder_with_skip_connection=True, d_out=64, out_hook_point_layer=6,
                          activation_fn_str="topk", activation_fn_kwargs={"k": 8}, use_ghost_grads=True, dead_feature_window=1),
    "tc_relu_ghost": dict(is_transcoder=True, transcoder_with_skip_connection=False, d_out=64, out_hook_point_layer=6,
                          activation_fn_str="relu", activation_fn_kwargs={}, l1_coefficient=2e-3, use_ghost_grads=True, dead_feature_window=1),
}

if __name__ == "__main__":
    blob = {}
    for v, over in VARIANTS.items():
        blob.update(run(v, over))
    np.savez_compressed(os.path.join(HERE, "sae_tail_steps.npz"), **blob)
    print("sae_tail_steps.npz", os.path.getsize(os.path.join(HERE, "sae_tail_steps.npz")) // 1024, "kB")
