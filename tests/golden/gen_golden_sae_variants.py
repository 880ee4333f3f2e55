"""Golden fixtures for the other sparse coders by EXECUTING THE REFERENCE (build container only):
the reference's StandardSparseAutoencoder with ReLU + L1 (+ ghost gradients) and with top-k + ghost gradients, GatedSparseAutoencoder
(ReLU and top-k forms) and Transcoder are
run through its own VisionSAETrainer.train_step (/root/reference/src/vit_prisma/sae/train_sae.py:278-411) for 3 steps at
d_in = 64, d_sae = 512, N = 256.

    python tests/golden/gen_golden_sae_variants.py     ->  tests/golden/sae_variants_steps.npz

Per variant and step: loss / mse / l1 / l0 / ghost / aux scalars, act_freq, n_since_fired; parameters after step 3.  The
initial parameters of every variant are generated here from a numpy RandomState and stored in the fixture too
(``<variant>_init_<name>``), so the test needs no knowledge of the reference's initialisers.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from gen_golden_sae import ref_trainer_classes  # noqa: E402
from vit_prisma_amd.synth import synth_sae_batch  # noqa: E402

D_IN, EXP, N = 64, 8, 256

VARIANTS = {
    "relu_l1": dict(activation_fn_str="relu", activation_fn_kwargs={}, l1_coefficient=2e-3, use_ghost_grads=False),
    "relu_ghost": dict(activation_fn_str="relu", activation_fn_kwargs={}, l1_coefficient=2e-3, use_ghost_grads=True,
                       dead_feature_window=1),
    "topk_ghost": dict(activation_fn_str="topk", activation_fn_kwargs={"k": 8}, use_ghost_grads=True, dead_feature_window=1),
    "gated": dict(architecture="gated", activation_fn_str="relu", activation_fn_kwargs={}, l1_coefficient=2e-3,
                  use_ghost_grads=False),
    "gated_topk": dict(architecture="gated", activation_fn_str="topk", activation_fn_kwargs={"k": 8}, use_ghost_grads=False),
    "transcoder_dout": dict(is_transcoder=True, transcoder_with_skip_connection=False, d_out=40, out_hook_point_layer=6,
                            activation_fn_str="topk", activation_fn_kwargs={"k": 8}, use_ghost_grads=False),
    "transcoder": dict(is_transcoder=True, transcoder_with_skip_connection=True, d_out=64, out_hook_point_layer=6,
                       activation_fn_str="topk", activation_fn_kwargs={"k": 8}, use_ghost_grads=False),
}


def make_cfg(Cfg, **over):
    kw = dict(hook_point_layer=6, layer_subtype="hook_resid_post", d_in=D_IN, expansion_factor=EXP, activation_fn_str="relu",
              activation_fn_kwargs={}, normalize_activations="layer_norm", initialization_method="independent",
              b_dec_init_method="mean", train_batch_size=N, lr=1e-3, max_grad_norm=1.0, _device="cpu", _dtype="float32",
              log_to_wandb=False, use_ghost_grads=False, feature_sampling_window=1000, dead_feature_window=5000,
              lr_scheduler_name="constant", n_checkpoints=0, verbose=False)
    kw.update(over)
    return Cfg(**kw)


def init_params(model, seed):
    rs = np.random.RandomState(seed)
    out = {}
    with torch.no_grad():
        for name, p in model.named_parameters():
            if p.ndim == 2:
                v = rs.uniform(-1.0, 1.0, size=tuple(p.shape)).astype(np.float32)
                v /= np.linalg.norm(v, axis=1, keepdims=True)
            else:
                v = (rs.standard_normal(size=tuple(p.shape)) * 0.05).astype(np.float32)
            p.copy_(torch.from_numpy(v))
            out[name] = v
    return out


class _PairActs:
    """layer_acts[:, 0, :] -> x, layer_acts[:, 1, :] -> y for tensors of different widths (what train_step indexes)."""

    def __init__(self, x, y):
        self.x, self.y = x, y
        self.shape = x.shape

    def __getitem__(self, idx):
        return (self.x, self.y)[idx[1]]


def run(variant, over):
    Cfg, SAE, Trainer = ref_trainer_classes()
    from vit_prisma.sae.sae import GatedSparseAutoencoder
    from vit_prisma.sae.transcoder import Transcoder
    cfg = make_cfg(Cfg, **over)
    torch.manual_seed(0)
    if over.get("is_transcoder"):
        model = Transcoder(cfg)
    elif over.get("architecture") == "gated":
        model = GatedSparseAutoencoder(cfg)
    else:
        model = SAE(cfg)
    blob = {f"{variant}_init_{n}": v for n, v in init_params(model, 11).items()}
    tr = object.__new__(Trainer)
    tr.cfg = cfg
    tr.is_transcoder = bool(over.get("is_transcoder", False))
    opt = torch.optim.Adam(model.parameters(), lr=cfg.lr)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=lambda s: 1.0)
    act, since, frac = torch.zeros(cfg.d_sae), torch.zeros(cfg.d_sae), 0
    if over.get("use_ghost_grads"):            # (relu_ghost, topk_ghost; round 6: the two transcoder ghost variants of gen_golden_sae_tail.py)
        since[::3] = 5.0                       # a third of the features count as dead (window 1): ghost grads are live
    blob[f"{variant}_since0"] = since.clone().numpy()
    for t in range(3):
        x = torch.from_numpy(synth_sae_batch(N, D_IN, seed=t))
        if tr.is_transcoder and over.get("d_out", D_IN) != D_IN:
            # a target of another width than the input (the first d_out columns of a batch of the usual kind): train_step only reads
            # layer_acts[:, 0] and [:, 1] (train_sae.py:299-301), a two-entry container per token serves
            y = torch.from_numpy(synth_sae_batch(N, D_IN, seed=100 + t)[:, :over["d_out"]].copy())
            layer_acts = _PairActs(x, y)
        elif tr.is_transcoder:
            y = torch.from_numpy(synth_sae_batch(N, D_IN, seed=100 + t))
            layer_acts = torch.stack([x, y], dim=1)
        else:
            layer_acts = x[:, None, :]
        captured = {}
        orig_fwd = model.forward

        def spy(*a, **k):
            out = orig_fwd(*a, **k)
            captured["ghost"], captured["aux"] = float(out[5]), float(out[6])
            return out

        model.forward = spy
        try:
            loss, mse, l1, l0, act, since, frac = tr.train_step(
                sparse_autoencoder=model, optimizer=opt, scheduler=sched, act_freq_scores=act, n_forward_passes_since_fired=since,
                n_frac_active_tokens=frac, layer_acts=layer_acts, n_training_steps=t, n_training_tokens=t * N)
        finally:
            model.forward = orig_fwd
        blob[f"{variant}_s{t}_scalars"] = np.array(
            [float(loss), float(mse), float(l1) if l1 is not None else np.nan, float(l0), captured["ghost"], captured["aux"]], np.float64)
        if t == 2:                                   # (parameters after the last step only: keeps the fixture small)
            for n, p in model.named_parameters():
                blob[f"{variant}_s{t}_param_{n}"] = p.detach().clone().numpy()
        blob[f"{variant}_s{t}_act_freq"] = act.clone().numpy()
        blob[f"{variant}_s{t}_n_since"] = since.clone().numpy()
        print(variant, t, blob[f"{variant}_s{t}_scalars"], flush=True)
    blob[f"{variant}_keys"] = np.array([n for n, _ in model.named_parameters()])
    return blob


if __name__ == "__main__":
    blob = {}
    for v, over in VARIANTS.items():
        blob.update(run(v, over))
    np.savez_compressed(os.path.join(HERE, "sae_variants_steps.npz"), **blob)
    print("sae_variants_steps.npz", os.path.getsize(os.path.join(HERE, "sae_variants_steps.npz")) // 1024, "kB")
