"""A few ReLU + L1 train steps (768 -> 24576, N = 4096) for rocprofv3 --kernel-trace --stats: pv_sae_relu_step in its sparse form (b_enc
shifted so that a token keeps ~16 features from the first step on: the regime the run from the synthetic init settles into), or
PV_RELU_L0=860 for the dense form at the published L0."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_prisma_amd.sae.bench_leg import sae_bench_leg
if os.environ.get("PV_TUNE"):                                     # A/B runs: "key=value[,key=value]"
    from vit_prisma_amd import _native
    for kv in os.environ["PV_TUNE"].split(","):
        _native.set_tuning(kv.split("=")[0], int(kv.split("=")[1]))
print(sae_bench_leg(torch.device("cuda", 0), steps=5, warmup=2, activation="relu", relu_target_l0=float(os.environ.get("PV_RELU_L0", "16"))))
