"""bench.py's two end-to-end legs in its order (store 256 x 8, then the reference config's 32 x 20) in ONE process, SAE step folds on or off
(argv[1] = sae_fold), optionally the step-only leg first (argv[2] = 1): which predecessor slows the second leg?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vit_prisma_amd import _native as N
from vit_prisma_amd.sae.bench_leg import sae_bench_leg, sae_end_to_end_leg

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
N.set_tuning("sae_fold", int(sys.argv[1]))
if len(sys.argv) > 2 and sys.argv[2] == "1":
    r = sae_bench_leg(dev, dist=None)
    print("step-only", r["ms_per_step"], flush=True)
    torch.cuda.empty_cache()
if not (len(sys.argv) > 3 and sys.argv[3] == "skip"):
    r = sae_end_to_end_leg(dev)
    print("e2e 256x8", r["ms_per_step"], flush=True)
    torch.cuda.empty_cache()
r = sae_end_to_end_leg(dev, steps=40, warmup=24, store_bs=32, n_buf=20)
print("e2e 32x20", r["ms_per_step"], flush=True)
