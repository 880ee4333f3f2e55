"""ReLU + L1, ghost gradients, Gated SAE and Transcoder on the PyTorch path of VisionSAETrainer,
against fixtures produced by running the REFERENCE's own classes through its own train_step
(tests/golden/gen_golden_sae_variants.py): three steps, scalars at 1e-5, parameters after step 3 at 1e-5, statistics exact."""
import os

import numpy as np
import pytest
import torch

from vit_prisma_amd.sae import (GatedSparseAutoencoder, StandardSparseAutoencoder, Transcoder, VisionModelSAERunnerConfig,
                                VisionSAETrainer)
from vit_prisma_amd.synth import synth_sae_batch

from conftest import GOLDEN, rel_fro

D_IN, EXP, N = 64, 8, 256
VARIANTS = {
    "relu_l1": dict(activation_fn_str="relu", activation_fn_kwargs={}, l1_coefficient=2e-3, use_ghost_grads=False),
    "relu_ghost": dict(activation_fn_str="relu", activation_fn_kwargs={}, l1_coefficient=2e-3, use_ghost_grads=True,
                       dead_feature_window=1),
    "gated": dict(architecture="gated", activation_fn_str="relu", activation_fn_kwargs={}, l1_coefficient=2e-3,
                  use_ghost_grads=False),
    "transcoder": dict(is_transcoder=True, transcoder_with_skip_connection=True, d_out=64, out_hook_point_layer=6,
                       activation_fn_str="topk", activation_fn_kwargs={"k": 8}, use_ghost_grads=False),
}


def make_cfg(**over):
    kw = dict(hook_point_layer=6, layer_subtype="hook_resid_post", d_in=D_IN, expansion_factor=EXP, activation_fn_str="relu",
              activation_fn_kwargs={}, normalize_activations="layer_norm", initialization_method="independent",
              b_dec_init_method="mean", train_batch_size=N, lr=1e-3, max_grad_norm=1.0, _device="cpu", _dtype="float32",
              log_to_wandb=False, use_ghost_grads=False, feature_sampling_window=1000, dead_feature_window=5000,
              lr_scheduler_name="constant", n_checkpoints=0, verbose=False)
    kw.update(over)
    return VisionModelSAERunnerConfig(**kw)


# round 6: the tail of 8(f) row 3 (tests/golden/gen_golden_sae_tail.py -> sae_tail_steps.npz)
TAIL = {
    "relu_constnorm": dict(activation_fn_str="relu", activation_fn_kwargs={}, l1_coefficient=2e-3, normalize_activations="constant_norm_rescale"),
    "topk_constnorm": dict(activation_fn_str="topk", activation_fn_kwargs={"k": 8}, normalize_activations="constant_norm_rescale"),
    "tanh_relu": dict(activation_fn_str="tanh-relu", activation_fn_kwargs={}, l1_coefficient=2e-3),
    "relu_lp2": dict(activation_fn_str="relu", activation_fn_kwargs={}, l1_coefficient=2e-3, lp_norm=2),
}


@pytest.mark.parametrize("variant", list(VARIANTS) + list(TAIL))
def test_variant_three_steps_match_the_reference(variant):
    g = np.load(os.path.join(GOLDEN, "sae_tail_steps.npz" if variant in TAIL else "sae_variants_steps.npz"))
    cfg = make_cfg(**{**VARIANTS, **TAIL}[variant])
    tr = VisionSAETrainer(cfg, model=None, dataset=None)          # the trainer picks the class (train_sae.py:72-81)
    model = tr.sparse_coder
    want_cls = {"gated": GatedSparseAutoencoder, "transcoder": Transcoder}.get(variant, StandardSparseAutoencoder)
    assert type(model) is want_cls
    keys = [str(k) for k in g[f"{variant}_keys"]]
    assert [n for n, _ in model.named_parameters()] == keys       # same parameters, same (state-dict) order
    with torch.no_grad():
        for n, p in model.named_parameters():
            p.copy_(torch.from_numpy(g[f"{variant}_init_{n}"]))
    act, since, frac, opt, sched = tr.initialize_training_variables()
    since.copy_(torch.from_numpy(g[f"{variant}_since0"]))
    for t in range(3):
        x = torch.from_numpy(synth_sae_batch(N, D_IN, seed=t))
        if variant == "transcoder":
            layer_acts = torch.stack([x, torch.from_numpy(synth_sae_batch(N, D_IN, seed=100 + t))], dim=1)
        else:
            layer_acts = x[:, None, :]
        loss, mse, l1, l0, act, since, frac = tr.train_step(
            sparse_autoencoder=model, optimizer=opt, scheduler=sched, act_freq_scores=act, n_forward_passes_since_fired=since,
            n_frac_active_tokens=frac, layer_acts=layer_acts, n_training_steps=t, n_training_tokens=t * N)
        assert not tr.last_step_native
        want = g[f"{variant}_s{t}_scalars"]
        assert abs(float(loss) - want[0]) <= 1e-5 * abs(want[0]), (t, float(loss), want[0])
        assert abs(float(mse) - want[1]) <= 1e-5 * abs(want[1])
        if np.isnan(want[2]):
            assert l1 is None
        else:
            assert abs(float(l1) - want[2]) <= 1e-5 * abs(want[2])
        assert abs(float(l0) - want[3]) <= 1e-6 * max(want[3], 1.0)
        assert np.array_equal(act.numpy(), g[f"{variant}_s{t}_act_freq"]) and np.array_equal(since.numpy(), g[f"{variant}_s{t}_n_since"])
    for n, p in model.named_parameters():
        assert rel_fro(p.detach().numpy(), g[f"{variant}_s2_param_{n}"]) < 1e-5, n
    # the 7-tuple contract
    out = model(x) if variant != "transcoder" else model(x, x)
    assert len(out) == 7 and out[0].shape == (N, D_IN) and out[1].shape == (N, D_IN * EXP)
