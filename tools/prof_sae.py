"""7 full SAE train steps (768 -> 24576, k = 32, N = 4096) for rocprofv3 --kernel-trace --stats / --pmc."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_prisma_amd.sae.bench_leg import sae_bench_leg
if os.environ.get("PV_TUNE"):                                     # A/B runs: "key=value[,key=value]"
    from vit_prisma_amd import _native
    for kv in os.environ["PV_TUNE"].split(","):
        _native.set_tuning(kv.split("=")[0], int(kv.split("=")[1]))
print(sae_bench_leg(torch.device("cuda", 0), steps=5, warmup=2))
