"""Build-container measurement: the REFERENCE's own classes -- HookedViT.run_with_cache and
VisionSAETrainer.train_step, imported from /root/reference through tests/golden/_refimport.py -- timed beside this package's
PyTorch CPU path (what bench.py's cpu_baseline times on the GPU box, where /root/reference does not exist) on the same cores with
the same inputs.  Writes profiles/r04_cpu_reference_vs_port.json.  Not runnable on the GPU box.

    python tools/cpu_reference_vs_port.py
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def median_time(fn, warm=2, reps=5):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)), ts


def vit_pair(bs=32):
    from gen_golden_vit import build_reference_model
    from vit_prisma_amd import HookedViT, HookedViTConfig
    from vit_prisma_amd.synth import ARCHS, synth_images, synth_vit_state
    arch = ARCHS["clip-vit-b32"]
    x = torch.from_numpy(synth_images(arch, bs, 1))
    ref, _ = build_reference_model("clip-vit-b32")
    port = HookedViT(HookedViTConfig(**arch, dtype=torch.float32, device="cpu"))
    port.load_state_dict({k: torch.from_numpy(v) for k, v in synth_vit_state(arch, 0).items()}, strict=True)
    port = port.eval().use_native(False)

    def run(model):
        def once():
            with torch.no_grad():
                _, cache = model.run_with_cache(x)
            assert len(cache) == 214
        return once

    t_ref, all_ref = median_time(run(ref))
    t_port, all_port = median_time(run(port))
    return {"workload": f"run_with_cache, all 214 hooks, CLIP ViT-B/32, bs={bs}, fp32", "reference_images_per_s": round(bs / t_ref, 2),
            "port_images_per_s": round(bs / t_port, 2), "port_over_reference": round(t_ref / t_port, 3),
            "reference_s": [round(t, 4) for t in all_ref], "port_s": [round(t, 4) for t in all_port]}


def sae_pair(d_in=768, exp=32, k=32, n=4096):
    from gen_golden_sae import make_cfg, ref_trainer_classes
    from vit_prisma_amd.sae import StandardSparseAutoencoder, VisionModelSAERunnerConfig, VisionSAETrainer
    from vit_prisma_amd.synth import synth_sae_batch, synth_sae_state
    Cfg, SAE, Trainer = ref_trainer_classes()
    sd = synth_sae_state(d_in, d_in * exp, 0)
    batches = [torch.from_numpy(synth_sae_batch(n, d_in, seed=t))[:, None, :] for t in range(2)]

    def build(CfgC, SAEC, TrainerC, ours):
        if ours:
            cfg = CfgC(hook_point_layer=6, layer_subtype="hook_resid_post", d_in=d_in, expansion_factor=exp, activation_fn_str="topk",
                       activation_fn_kwargs={"k": k}, normalize_activations="layer_norm", initialization_method="independent",
                       b_dec_init_method="mean", train_batch_size=n, lr=1e-3, max_grad_norm=1.0, _device="cpu", _dtype="float32",
                       log_to_wandb=False, lr_scheduler_name="constant", n_checkpoints=0, verbose=False)
        else:
            cfg = make_cfg(CfgC, d_in, exp, k, n)
        sae = SAEC(cfg)
        with torch.no_grad():
            for name, val in sd.items():
                getattr(sae, name).copy_(torch.from_numpy(val))
        if ours:
            tr = TrainerC(cfg, model=None, dataset=None, sparse_coder=sae).use_native(False)
        else:
            tr = object.__new__(TrainerC)                 # (its __init__ builds a model + an activation store)
            tr.cfg = cfg
            tr.is_transcoder = False
        opt = torch.optim.Adam(sae.parameters(), lr=cfg.lr)
        sched = torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=lambda s: 1.0)
        state = {"act": torch.zeros(cfg.d_sae), "since": torch.zeros(cfg.d_sae), "frac": 0, "t": 0}

        def once():
            t = state["t"]
            out = tr.train_step(sparse_autoencoder=sae, optimizer=opt, scheduler=sched, act_freq_scores=state["act"],
                                n_forward_passes_since_fired=state["since"], n_frac_active_tokens=state["frac"],
                                layer_acts=batches[t % 2], n_training_steps=t, n_training_tokens=t * n)
            state["act"], state["since"], state["frac"] = out[4], out[5], out[6]
            state["t"] = t + 1
        return once

    t_ref, all_ref = median_time(build(Cfg, SAE, Trainer, False), warm=1, reps=5)
    t_port, all_port = median_time(build(VisionModelSAERunnerConfig, StandardSparseAutoencoder, VisionSAETrainer, True), warm=1, reps=5)
    return {"workload": f"VisionSAETrainer.train_step, top-k SAE {d_in} -> {d_in * exp}, k={k}, {n} tokens, fp32, PyTorch autograd",
            "reference_tokens_per_s": round(n / t_ref, 1), "port_tokens_per_s": round(n / t_port, 1),
            "port_over_reference": round(t_ref / t_port, 3), "reference_s": [round(t, 3) for t in all_ref],
            "port_s": [round(t, 3) for t in all_port]}


if __name__ == "__main__":
    torch.set_num_threads(os.cpu_count() or 1)
    res = {"host": {"cores": os.cpu_count(), "torch_threads": torch.get_num_threads(), "torch": torch.__version__},
           "note": "build container (no GPU): the reference's own classes next to this package's PyTorch CPU path -- the path bench.py's "
                   "cpu_baseline ('kind': 'port') times on the GPU box, where the reference tree does not exist",
           "vit": vit_pair(), "sae": sae_pair()}
    out = os.path.join(ROOT, "profiles", "r04_cpu_reference_vs_port.json")
    with open(out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res, indent=1))
