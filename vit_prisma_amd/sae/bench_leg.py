"""SAE half of the headline metric: train-step tokens/s (BASELINE.json configs[2] / [3]).

Step-only measurement: batches are pre-staged in HBM (``randn(4096, 768) * 3 + per-dim offset``), a
step is the full reference ``train_step`` (renorm decoder -> forward -> stats -> backward -> clip ->
project -> Adam), fp32 master weights, d_in = 768, d_sae = 24576 (32x), top-k = 32.
With a process group the GLOBAL batch stays 4096 tokens (strong scaling).  Default mode: feature parallel (each rank owns
d_sae / W features for good and sees all 4096 tokens; token-sized collectives only -- sae/feature_parallel.py);
``mode="data"``: each rank takes 4096 / W tokens and 1 / W of the optimizer (sae/trainer.py: reduce-scatter of gradient
rows, all-gather of parameters).  ``weak=True``: 4096 tokens PER RANK (global batch 4096 W), the data-parallel mode.
"""
from __future__ import annotations

import time
from typing import Optional

import numpy as np
import torch

from .. import _native as N
from ..synth import synth_sae_batch, synth_sae_state
from .native_sae import NativeSAE

D_IN, D_SAE, TOPK, N_TOKENS = 768, 24576, 32, 4096
PEAK_HBM_GBS = 8000.0
PEAK_F32_TFLOPS = 157.3
PEAK_F16_TFLOPS = 2500.0                     # dense fp16 / bf16 MFMA (MI355X_MICROARCH.md: ~2.5 PF dense)


def _pmc_step_traffic() -> dict:
    """HBM bytes of ONE train step from the newest committed rocprofv3 PMC summary of tools/prof_sae.py
    (profiles/*pmc_traffic_sae*.json: separate FETCH_SIZE / WRITE_SIZE passes over 7 steps, gfx950 2x FETCH correction,
    summed over every kernel of the step)."""
    import glob
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    paths = sorted(glob.glob(os.path.join(root, "profiles", "r*_pmc_traffic_sae*.json")))      # newest round last, by name
    for path in reversed(paths):
        try:
            with open(path) as f:
                d = json.load(f)
            ker = d["kernels"]
            steps = max(v["launches"] for k, v in ker.items() if "sae_decode_kernel" in k)
            total = sum(v["hbm_bytes_per_launch_corrected"] * v["launches"] for v in ker.values())
            return {"traffic": int(total / steps), "traffic_source": f"{os.path.basename(path)} (all kernels of {steps} profiled steps)"}
        except Exception:
            continue
    return {"traffic": None}


def sae_bench_leg(dev: torch.device, dist=None, steps: int = 20, warmup: int = 5, feature_parallel: bool = True,
                  weak: bool = False, activation: str = "topk", relu_target_l0: Optional[float] = None) -> dict:
    """Step-only: ``VisionSAETrainer.train_step`` (the reference's call, train_sae.py:278-411) on batches resident in HBM.
    With a process group: feature_parallel (default) = the feature-sharded step of sae/feature_parallel.py (tokens / candidates
    all-gathered, partial reconstructions all-reduced; no gradient or parameter traffic); otherwise every rank takes 4096 / W
    tokens of the same global batch and the trainer's sharded-optimizer step runs (reduce-scatter of gradient rows, local
    clip / project / Adam on 1 / W of the features, asynchronous all-gather of the parameters).  weak: 4096 tokens per rank
    (data parallel).  activation "relu": the ReLU + L1 SAE (l1_coefficient 8e-5) instead of top-k, on pv_sae_relu_step -- sparse where
    the batch allows it, the dense GEMMs otherwise, decided on the GPU; the leg counts the steps of either kind.  From the synthetic
    init the first ~5 steps are dense (half of all features fire), then L0 collapses under the L1 term and the step runs sparse:
    ``warmup`` decides which regime is timed.  relu_target_l0: shift b_enc so that a token keeps about that many features from the
    first step on (e.g. 0.035 * d_sae: the L0 of the reference's published x64 SAEs, docs/sae_table.md:12-36) and stop the encoder
    bias from training the regime away (lr 0: the step's arithmetic and traffic are those of training, the state stays put)."""
    from .config import VisionModelSAERunnerConfig
    from .sae import StandardSparseAutoencoder
    from .trainer import VisionSAETrainer
    world = dist.get_world_size() if dist is not None else 1
    rank = dist.get_rank() if dist is not None else 0
    relu = activation == "relu"
    feature_parallel = bool(feature_parallel) and world > 1 and not weak and not relu
    n_global = N_TOKENS * world if weak else N_TOKENS
    n_local = N_TOKENS if weak else N_TOKENS // world
    cfg = VisionModelSAERunnerConfig(
        hook_point_layer=6, layer_subtype="hook_resid_post", d_in=D_IN, expansion_factor=D_SAE // D_IN,
        activation_fn_str="relu" if relu else "topk", activation_fn_kwargs={} if relu else {"k": TOPK},
        normalize_activations="layer_norm", l1_coefficient=8e-5,
        initialization_method="independent", b_dec_init_method="mean", train_batch_size=n_global, lr=1e-3,
        max_grad_norm=1.0, _device=str(dev), log_to_wandb=False, lr_scheduler_name="constant", n_checkpoints=0)
    sae = StandardSparseAutoencoder(cfg)
    with torch.no_grad():
        for n, v in synth_sae_state(D_IN, D_SAE, 0).items():
            getattr(sae, n).copy_(torch.from_numpy(v))
    if relu and relu_target_l0 is not None:
        with torch.no_grad():
            xb = torch.from_numpy(synth_sae_batch(512, D_IN, seed=0)).to(dev)
            xh = (xb - xb.mean(-1, keepdim=True)) / (xb.std(-1, keepdim=True) + 1e-5)
            h = (xh - sae.b_dec) @ sae.W_enc + sae.b_enc
            sae.b_enc -= torch.quantile(h.flatten()[::97].float(), 1.0 - float(relu_target_l0) / D_SAE)
            del xb, xh, h
        cfg.lr = 0.0
    tr = VisionSAETrainer(cfg, model=None, dataset=None, sparse_coder=sae).use_native(True)
    tr.use_feature_parallel(feature_parallel)
    st = list(tr.initialize_training_variables())                 # act_freq, n_since_fired, n_frac, optimizer, scheduler
    n_dense = torch.zeros(1, dtype=torch.int32, device=dev)       # steps of the timed region that ran on the dense GEMMs (device-side count)
    batches = [torch.from_numpy(synth_sae_batch(n_global, D_IN, seed=i)).to(dev)[rank * n_local:(rank + 1) * n_local][:, None, :].contiguous()
               for i in range(4)]
    n_done = [0]

    def step(x: torch.Tensor) -> None:
        _, _, _, _, st[0], st[1], st[2] = tr.train_step(
            sparse_autoencoder=sae, optimizer=st[3], scheduler=st[4], act_freq_scores=st[0], n_forward_passes_since_fired=st[1],
            n_frac_active_tokens=st[2], layer_acts=x, n_training_steps=n_done[0], n_training_tokens=n_done[0] * n_global)
        n_done[0] += 1
        if relu and getattr(tr._engine, "_relu_ws", None) is not None:
            n_dense.add_(tr._engine.relu_mode)                    # (asynchronous: a device word added to a device counter)

    for i in range(warmup):
        step(batches[i % 4])
    n_dense.zero_()
    torch.cuda.synchronize(dev)
    N.prof_reset()
    N.prof_enable(True)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(steps):
        step(batches[i % 4])
    tr._dp_flush()                                                # the last step's parameters have landed inside the timed region
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    N.prof_enable(False)
    tr.sync_parameters()                                          # (outside the timed region: module parameters complete and current)
    eng = tr._fp.engine if feature_parallel else tr._engine
    assert tr.last_step_native and eng is not None
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    loss = float((tr._fp.loss if feature_parallel else eng.scalars[0]).item())
    enc = N.prof_read("sae_encode_topk")
    bwd = N.prof_read("sae_backward")
    app = N.prof_read("sae_apply")
    ms_step = elapsed / steps * 1e3
    if relu:
        dense_steps = int(n_dense.item())
        sparse_steps = steps - dense_steps
        # dense steps: five fp32 MFMA GEMMs of 2 N d_in d_sae FLOP each (sae_dense.hip), bound by the fp32 matrix peak; sparse steps:
        # the top-k step's traffic (1.4 GB of algorithmic HBM bytes), bound by HBM
        if dense_steps == steps:
            # since round 6 the five GEMMs run on the fp16 matrix pipe as three fp16 products per fp32 product (split-fp16, sae_dense.hip):
            # the roof is the dense fp16 MFMA peak over the 30 N d_in d_sae fp16 FLOP issued; the algorithmic (fp32-equivalent) rate
            # and its fraction of the fp32 matrix peak -- the roof of round 5's form -- stand beside it
            flops = 10.0 * n_local * D_IN * D_SAE
            tf = flops / (ms_step * 1e-3) / 1e12
            split = N.get_tuning("dense_fp32") == 0
            if split:
                roof = {"kernel": "whole step vs 30 N d_in d_sae fp16 FLOP (five dense GEMMs, three fp16 products per fp32 product, fp32 accumulation)",
                        "bound": "mfma", "achieved": round(3.0 * tf, 1), "peak": PEAK_F16_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(3.0 * tf / PEAK_F16_TFLOPS, 4), "algorithmic_TFLOPs_fp32_equivalent": round(tf, 1),
                        "frac_of_fp32_matrix_peak": round(tf / PEAK_F32_TFLOPS, 4)}
            else:
                roof = {"kernel": "whole step vs 10 N d_in d_sae FLOP (five dense fp32 GEMMs)", "bound": "mfma", "achieved": round(tf, 1),
                        "peak": PEAK_F32_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / PEAK_F32_TFLOPS, 4)}
        elif dense_steps == 0:
            gbs = 1.4e9 / (ms_step * 1e-3) / 1e9
            roof = {"kernel": "whole step vs the 1.4 GB of algorithmic HBM bytes of a k-sparse step (SURVEY.md 8d)", "bound": "hbm",
                    "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4)}
        else:
            roof = {"kernel": "mixed: some steps dense (MFMA-bound), some sparse (HBM-bound) -- no single roof", "bound": "n/a",
                    "achieved": None, "peak": None, "unit": None, "frac": None}
        return {
            "metric": "SAE train-step tokens/sec, ReLU + L1 SAE (pv_sae_relu_step: sparse where the batch allows, batches resident in HBM)",
            "value": round(n_global * steps / elapsed, 1), "unit": "tokens/s", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": round(ms_step, 3), "dtype": "f32",
            "config": {"workload": f"ReLU + L1 SAE 768 -> 24576 (32x), l1_coefficient 8e-5, {n_global} tokens per step, Adam, clip 1.0",
                       "regime": (f"b_enc shifted to L0 ~ {relu_target_l0:g} per token, lr 0 (state held)" if relu_target_l0 is not None else
                                  f"training run from the synthetic init, steps {warmup + 1}..{warmup + steps} timed (the first ~5 steps of "
                                  "this run are dense: half of all features fire; the L1 term then collapses L0 to a few features)"),
                       "arithmetic": "sparse steps: fp16 MFMA filter at the threshold -B_n + exact fp32 re-scoring, k-sparse kernels; dense "
                                     "steps: split-fp16 MFMA GEMMs (hi / lo fp16 halves of each fp32 operand, three v_mfma_f32_32x32x16_f16 "
                                     "products, fp32 accumulation: within 8e-7 of a float64 run, profiles/r06_dense_split_err.json)",
                       "per_token_capacity": int(getattr(eng, "relu_cap", 0))},
            "final_loss": loss, "l0": float(eng.scalars[2].item()), "sparse_steps": sparse_steps, "dense_steps": dense_steps,
            "roofline": roof,
            "kernels": {"encoder": {"avg_us": round(enc["ms"] * 1e3 / max(enc["launches"], 1), 1)},
                        "decoder_and_backward": {"avg_us": round(bwd["ms"] * 1e3 / max(bwd["launches"], 1), 1)},
                        "clip_project_adam": {"avg_us": round(app["ms"] * 1e3 / max(app["launches"], 1), 1)}},
        }
    # algorithmic HBM bytes of one step, SURVEY.md 8(d): Adam 7 x 151.1 MB = 1.06 GB + W_enc / W_dec reads for forward and
    # backward 0.30 GB + x / out 25 MB ~ 1.4 GB (the figure the roofline fraction is quoted against).  NOMINAL: the same
    # single-process figure whatever the world size (a rank of W moves about 1 / W of it plus the collectives' bytes)
    alg_bytes = 1.4e9
    res = {
        "metric": "SAE train-step tokens/sec (step-only, batches resident in HBM)",
        "value": round(n_global * steps / elapsed, 1), "unit": "tokens/s", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": round(ms_step, 3), "scaling": ("weak" if weak else "strong") if world > 1 else "n/a", "dtype": "f32",
        "config": {"workload": f"top-k SAE 768 -> 24576 (32x), k=32, global batch {n_global} tokens, Adam, clip 1.0",
                   "tokens_per_gpu_per_step": n_local,
                   "parallelism": "single process" if world == 1 else
                   (f"tp{world}: features sharded for good, tokens / candidates all-gathered, partial reconstructions all-reduced"
                    if feature_parallel else
                    f"dp{world}: tokens sharded, optimizer sharded by feature (reduce-scatter grads, all-gather params)"),
                   "encoder": "fp16 MFMA filter + exact fp32 re-scoring" if eng.filtered_encoder else "exact fp32 MFMA"},
        "final_loss": loss,
        "roofline": {"kernel": "whole step (every kernel of one train step) vs the 1.4 GB of algorithmic HBM bytes per step of SURVEY.md 8(d)",
                     "bound": "hbm", "achieved": round(alg_bytes / (ms_step * 1e-3) / 1e9, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                     "frac": round(alg_bytes / (ms_step * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
                     "bytes_model": "nominal: the single-process 1.4 GB per 4096-token step at every world size",
                     **(_pmc_step_traffic() if world == 1 else {"traffic": None})},
        "kernels": {
            "encode_topk": {"avg_us": round(enc["ms"] * 1e3 / max(enc["launches"], 1), 1),
                            "algorithmic_TFLOPs": round(enc["flops"] / max(enc["ms"], 1e-9) / 1e9, 1),
                            "fallback_rows_last_step": eng.fallback_rows()},
            "decode_csr_backward": {"avg_us": round(bwd["ms"] * 1e3 / max(bwd["launches"], 1), 1)},
            "clip_project_adam": {"avg_us": round(app["ms"] * 1e3 / max(app["launches"], 1), 1),
                                  "GBps": round(app["bytes"] / max(app["ms"], 1e-9) / 1e6, 1)},
        },
    }
    return res


def sae_variants_leg(dev: torch.device, steps: int = 8, warmup: int = 3, only: Optional[str] = None) -> dict:
    """Step-only times of the other coders of the reference on their fused HIP steps, through ``VisionSAETrainer.train_step`` at
    the bench shape (768 -> 24576, 4096 tokens, single process): a top-k Transcoder with the skip connection (sae/transcoder.py)
    and a Gated SAE with the ReLU magnitude path (sae.py:648-792) -- as a training run from the synthetic init (every gate half open:
    the step's dense form) and with b_gate shifted so that a token opens about 64 gates, lr 0 (a trained gated SAE's regime: the sparse
    form, pv_sae_gated_step_sparse; which form ran is decided on the GPU and counted) -- and a Gated SAE in its top-k form
    (pv_sae_gated_topk_step)."""
    from .config import VisionModelSAERunnerConfig
    from .trainer import VisionSAETrainer
    out = {}
    for name, over in (("transcoder_topk_skip", dict(activation_fn_str="topk", activation_fn_kwargs={"k": TOPK}, is_transcoder=True,
                                                     transcoder_with_skip_connection=True, d_out=D_IN, out_hook_point_layer=6)),
                       ("gated_relu", dict(activation_fn_str="relu", activation_fn_kwargs={}, architecture="gated", l1_coefficient=8e-5)),
                       ("gated_relu_l0_64", dict(activation_fn_str="relu", activation_fn_kwargs={}, architecture="gated",
                                                 l1_coefficient=8e-5)),
                       ("gated_topk", dict(activation_fn_str="topk", activation_fn_kwargs={"k": TOPK}, architecture="gated"))):
        if only is not None and name != only:
            continue
        target_l0 = 64.0 if name == "gated_relu_l0_64" else None
        cfg = VisionModelSAERunnerConfig(
            hook_point_layer=6, layer_subtype="hook_resid_post", d_in=D_IN, expansion_factor=D_SAE // D_IN,
            normalize_activations="layer_norm", initialization_method="independent", b_dec_init_method="mean",
            train_batch_size=N_TOKENS, lr=0.0 if target_l0 else 1e-3, max_grad_norm=1.0, _device=str(dev), log_to_wandb=False,
            lr_scheduler_name="constant", n_checkpoints=0, **over)
        tr = VisionSAETrainer(cfg, model=None, dataset=None).use_native(True)
        sae = tr.sparse_coder
        with torch.no_grad():
            for n, v in synth_sae_state(D_IN, D_SAE, 0).items():
                getattr(sae, n).copy_(torch.from_numpy(v))
            if target_l0:
                xb = torch.from_numpy(synth_sae_batch(512, D_IN, seed=0)).to(dev)
                xh = (xb - xb.mean(-1, keepdim=True)) / (xb.std(-1, keepdim=True) + 1e-5)
                h = (xh - sae.b_dec) @ sae.W_enc + sae.b_gate
                sae.b_gate -= torch.quantile(h.flatten()[::97].float(), 1.0 - target_l0 / D_SAE)
                del xb, xh, h
        gated = over.get("architecture") == "gated"
        n_dense = torch.zeros(1, dtype=torch.int32, device=dev)
        st = list(tr.initialize_training_variables())
        pair = cfg.is_transcoder
        xs = [torch.from_numpy(synth_sae_batch(N_TOKENS, D_IN, seed=i)).to(dev) for i in range(4)]
        if pair:
            xs = [torch.stack([x, torch.from_numpy(synth_sae_batch(N_TOKENS, D_IN, seed=50 + i)).to(dev)], dim=1).contiguous()
                  for i, x in enumerate(xs)]
        else:
            xs = [x[:, None, :].contiguous() for x in xs]
        n_done = [0]
        last = [None]

        def step(x):
            r = tr.train_step(sparse_autoencoder=sae, optimizer=st[3], scheduler=st[4], act_freq_scores=st[0],
                              n_forward_passes_since_fired=st[1], n_frac_active_tokens=st[2], layer_acts=x, n_training_steps=n_done[0],
                              n_training_tokens=n_done[0] * N_TOKENS)
            last[0], st[0], st[1], st[2] = r[0], r[4], r[5], r[6]
            n_done[0] += 1
            if gated and getattr(tr._engine, "_gated_ws", None) is not None:
                n_dense.add_(tr._engine._gated_ws[:4].view(torch.int32))      # (the step's mode word: a device word added to a device counter)

        for i in range(warmup):
            step(xs[i % 4])
        n_dense.zero_()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for i in range(steps):
            step(xs[i % 4])
        torch.cuda.synchronize(dev)
        elapsed = time.perf_counter() - t0
        assert tr.last_step_native
        out[name] = {"value": round(N_TOKENS * steps / elapsed, 1), "unit": "tokens/s", "ms_per_step": round(elapsed / steps * 1e3, 3),
                     "steps": steps, "warmup": warmup, "final_loss": float(last[0]),
                     "config": {"workload": f"{name}: 768 -> 24576, {N_TOKENS} tokens per step, Adam, clip 1.0, fused HIP step"}}
        if gated and name == "gated_topk":
            out[name].update({"l0": float(tr._engine.scalars[2].item())})
            out[name]["config"]["regime"] = (f"top-k form (TopK k = {TOPK} on the magnitudes and on the gate activations): two k-sparse lists per "
                                             f"token, training run from the synthetic init, steps {warmup + 1}..{warmup + steps} timed")
        elif gated:
            eng = tr._engine
            out[name].update({"l0": float(eng.scalars[2].item()), "dense_steps": int(n_dense.item()),
                              "sparse_steps": steps - int(n_dense.item()), "per_token_capacity": int(eng.relu_cap)})
            out[name]["config"]["regime"] = (f"b_gate shifted to ~{target_l0:g} open gates per token, lr 0 (state held)" if target_l0 else
                                             f"training run from the synthetic init, steps {warmup + 1}..{warmup + steps} timed")
        del tr, sae, xs
        torch.cuda.empty_cache()
    return out


class _ResidentImages(torch.utils.data.Dataset):
    """Synthetic image set that lives in HBM: items are (image view, label) like the reference's datasets."""

    def __init__(self, n: int, dev: torch.device, dtype: torch.dtype, seed: int):
        g = torch.Generator(device=dev).manual_seed(seed)
        self.x = torch.randn(n, 3, 224, 224, device=dev, generator=g).to(dtype)

    def __len__(self) -> int:
        return self.x.shape[0]

    def __getitem__(self, i):
        return self.x[i], 0


def sae_end_to_end_leg(dev: torch.device, dist=None, steps: int = 52, warmup: int = 64, feature_parallel: bool = True,
                       overlap_harvest: bool = True, store_bs: int = 256, n_buf: int = 8) -> dict:
    """Config 3, second number: the training loop the reference runs -- VisionActivationsStore
    harvesting ``blocks.6.hook_resid_post`` from randn images through ViT blocks 0..6 (native run_with_cache,
    names_filter + stop_at_layer), half-buffer shuffle-mix, VisionSAETrainer.train_step on the fused native step.
    ViT weights / activations bf16, SAE master weights fp32.  One-time work (first buffer fill, b_dec init, plan
    and engine creation) is outside the timed region; buffer refills are inside it.  The half-buffer mix serves
    76 800, 64 000, 57 600, ... -> 51 200 tokens per refill of 51 200 harvested ones, so the default warm-up (64 steps =
    four refills) runs past that start-up transient and the timed region (52 steps) covers four refill cycles at the
    steady state of one harvested token per trained token; the measured ratio is reported."""
    from .. import HookedViT, HookedViTConfig
    from ..synth import ARCHS, synth_vit_state
    from .config import VisionModelSAERunnerConfig
    from .sae import StandardSparseAutoencoder
    from .trainer import VisionSAETrainer

    world = dist.get_world_size() if dist is not None else 1
    rank = dist.get_rank() if dist is not None else 0
    arch = ARCHS["clip-vit-b32"]
    model = HookedViT(HookedViTConfig(**arch, dtype=torch.bfloat16, device="cuda"))
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth_vit_state(arch, 0).items()}, strict=True)
    model = model.to(torch.bfloat16).to(dev).eval().use_native(True)
    # store shape: 256 x 8 by default (the GPU wants big harvest batches); (32, 20) = the reference config's own defaults
    # (config.py:351-352), timed as bench.py's sae.end_to_end_reference_store_shape
    cfg = VisionModelSAERunnerConfig(
        hook_point_layer=6, layer_subtype="hook_resid_post", d_in=D_IN, expansion_factor=D_SAE // D_IN,
        activation_fn_str="topk", activation_fn_kwargs={"k": TOPK}, normalize_activations="layer_norm",
        initialization_method="independent", b_dec_init_method="mean", train_batch_size=N_TOKENS, lr=1e-3,
        max_grad_norm=1.0, _device=str(dev), log_to_wandb=False, lr_scheduler_name="constant", n_checkpoints=0,
        context_size=50, store_batch_size=store_bs, n_batches_in_buffer=n_buf)
    # every rank holds the same index space; the store's DistributedSampler hands each rank 4 store batches per epoch
    data = _ResidentImages(max(4 * store_bs, 256) * world, dev, torch.bfloat16, seed=77)
    sae = StandardSparseAutoencoder(cfg)
    tr = VisionSAETrainer(cfg, model=model, dataset=data, sparse_coder=sae).use_feature_parallel(bool(feature_parallel))
    act, since, frac, opt, sched = tr.initialize_training_variables()
    tr.initialize_geometric_medians()
    store = tr.activations_store
    store.overlap_harvest = bool(overlap_harvest)         # the next refill's ViT forwards on a side stream, behind the train steps
    n_steps = 0

    def one():
        nonlocal act, since, frac, n_steps
        x = store.next_batch()
        _, _, _, _, act, since, frac = tr.train_step(
            sparse_autoencoder=sae, optimizer=opt, scheduler=sched, act_freq_scores=act,
            n_forward_passes_since_fired=since, n_frac_active_tokens=frac, layer_acts=x,
            n_training_steps=n_steps, n_training_tokens=n_steps * N_TOKENS)
        n_steps += 1
        return x.shape[0]

    for _ in range(warmup):
        one()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    harvested0 = store.n_tokens_harvested
    t0 = time.perf_counter()
    tokens = 0
    for _ in range(steps):
        tokens += one()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert model.last_run_native, model.native_fallback_reason
    return {
        "metric": "SAE training tokens/sec end to end (harvest blocks 0..6 + shuffle buffer + train step)",
        "value": round(tokens * world / elapsed, 1), "unit": "tokens/s", "n_gpus": world, "steps": steps,
        "ms_per_step": round(elapsed / steps * 1e3, 3), "dtype": "bf16 ViT / f32 SAE",
        "config": {"workload": f"VisionSAETrainer loop: store_batch_size {store_bs} x n_batches_in_buffer {n_buf}, "
                               f"global train batch {N_TOKENS} tokens, hook blocks.6.hook_resid_post, images resident in HBM",
                   "tokens_per_gpu_per_step": N_TOKENS // world,
                   "harvest": "next refill prefetched on a side stream (overlaps the train steps)" if overlap_harvest else "synchronous",
                   "parallelism": "single process" if world == 1 else
                   ("images sharded for the harvest, features sharded for the step (tokens all-gathered)" if feature_parallel
                    else "images and tokens sharded, optimizer sharded by feature")},
        "flop_per_token": {"harvest": 104.8e6, "sae_step": 37.95e6}, "warmup": warmup,
        "harvested_tokens_per_trained_token": round((store.n_tokens_harvested - harvested0) / max(tokens, 1), 3),
    }
