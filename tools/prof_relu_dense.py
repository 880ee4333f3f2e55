"""rocprofv3 target: the ReLU + L1 step at the L0 of the reference's published SAEs (0.035 * d_sae ~ 860 features per token: the dense
GEMMs, since round 6 on the split-fp16 matrix path), 768 -> 24576, 4096 tokens; STEPS timed steps after 2 warm-up steps."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vit_prisma_amd.sae.bench_leg import D_SAE, sae_bench_leg

if os.environ.get("PV_TUNE"):                                     # A/B runs: "key=value[,key=value]"
    from vit_prisma_amd import _native
    for kv in os.environ["PV_TUNE"].split(","):
        _native.set_tuning(kv.split("=")[0], int(kv.split("=")[1]))
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
r = sae_bench_leg(dev, dist=None, steps=int(os.environ.get("STEPS", "7")), warmup=2, activation="relu", relu_target_l0=0.035 * D_SAE)
print(json.dumps({k: r.get(k) for k in ("value", "unit", "ms_per_step", "l0", "dense_steps", "sparse_steps", "final_loss", "roofline")}))
