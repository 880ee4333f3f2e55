"""A/B of the dense SAE steps on the split-fp16 matrix path (round 6) against the exact fp32 one (tuning key dense_fp32): the ReLU + L1
step at the published L0 (0.035 * d_sae) and from initialisation, and the gated step from initialisation; step times + losses."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vit_prisma_amd import _native as N
from vit_prisma_amd.sae.bench_leg import D_SAE, sae_bench_leg, sae_variants_leg

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
out = {}
for fp32 in (1, 0):
    N.set_tuning("reset")
    N.set_tuning("dense_fp32", fp32)
    tag = "fp32" if fp32 else "split_fp16"
    r = sae_bench_leg(dev, dist=None, steps=8, warmup=2, activation="relu", relu_target_l0=0.035 * D_SAE)
    out[f"relu_published_l0_{tag}"] = {k: r.get(k) for k in ("ms_per_step", "final_loss", "l0", "dense_steps")}
    print(tag, "relu published l0", out[f"relu_published_l0_{tag}"], flush=True)
    r = sae_bench_leg(dev, dist=None, steps=4, warmup=0, activation="relu")
    out[f"relu_from_init_{tag}"] = {k: r.get(k) for k in ("ms_per_step", "final_loss", "l0", "dense_steps")}
    print(tag, "relu from init", out[f"relu_from_init_{tag}"], flush=True)
    if os.environ.get("GATED", "1") == "1":
        r = sae_variants_leg(dev, steps=6, warmup=2, only="gated_relu")
        out[f"gated_{tag}"] = r
        print(tag, "gated", json.dumps(r)[:400], flush=True)
N.set_tuning("reset")
print(json.dumps(out))
