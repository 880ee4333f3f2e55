"""A/B of pv_sae_step's side stream (tuning key sae_side): the top-k step with the batch mean / weight bound beside prep + the sample GEMM
and the CSR build beside the decode kernel, against every launch on the caller's stream.  Alternating legs on one box; step time + loss."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vit_prisma_amd import _native as N
from vit_prisma_amd.sae.bench_leg import sae_bench_leg

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
rows = []
MODES = tuple(int(m) for m in os.environ.get("MODES", "0,1,2,3").split(","))   # 2 / 3: the pre-pass fork / the CSR fork alone
for rep in range(int(os.environ.get("REPS", "3"))):
    for side in MODES:
        N.set_tuning("reset")
        N.set_tuning("sae_side", side)
        r = sae_bench_leg(dev, dist=None, steps=int(os.environ.get("STEPS", "40")), warmup=5)
        rows.append({"sae_side": side, "ms_per_step": r["ms_per_step"], "final_loss": r.get("final_loss"), "l0": r.get("l0")})
        print(rows[-1], flush=True)
N.set_tuning("reset")
for side in MODES:
    v = sorted(x["ms_per_step"] for x in rows if x["sae_side"] == side)
    print(f"sae_side={side}: median {v[len(v) // 2]:.4f} ms  (min {v[0]:.4f}, max {v[-1]:.4f})")
print(json.dumps(rows))
