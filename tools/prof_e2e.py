"""The end-to-end SAE training leg of bench.py alone (for rocprofv3 --kernel-trace --stats)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vit_prisma_amd.sae.bench_leg import sae_end_to_end_leg
r = sae_end_to_end_leg(torch.device("cuda:0"), dist=None)
print(json.dumps({k: r[k] for k in ("value", "ms_per_step", "steps", "warmup") if k in r}))
