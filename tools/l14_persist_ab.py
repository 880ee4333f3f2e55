"""L/14@336 pattern leg with the persistent GEMM (gemm_persist: -1 auto, 0 never, 1 wherever it applies), alternating runs."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from vit_prisma_amd import _native as N

dev = torch.device("cuda:0")
res = {}
for rnd in range(2):
    for persist in (0, -1):
        N.set_tuning("reset")
        N.set_tuning("gemm_persist", persist)
        r = bench.l14_pattern_leg(dev, None, steps=int(os.environ.get("STEPS", "6")))
        res.setdefault(str(persist), []).append((r["value"], r["ms_per_step"]))
        print(persist, r["value"], r["ms_per_step"], flush=True)
N.set_tuning("reset")
print(json.dumps(res))
