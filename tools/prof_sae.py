"""rocprofv3 target: a few SAE train steps at BASELINE config 3."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vit_prisma_amd.sae.bench_leg import sae_bench_leg
print(sae_bench_leg(torch.device("cuda:0"), steps=5, warmup=2))
