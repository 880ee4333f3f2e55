"""A few ReLU + L1 dense train steps (768 -> 24576, N = 4096) for rocprofv3 --kernel-trace --stats."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_prisma_amd.sae.bench_leg import sae_bench_leg
print(sae_bench_leg(torch.device("cuda", 0), steps=4, warmup=2, activation="relu"))
