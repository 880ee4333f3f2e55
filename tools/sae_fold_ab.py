"""A/B of the folded launches of the top-k SAE step (tuning key sae_fold, default 1: the pre-pass in three launches instead of six, the CSR scan
as a role of the decode launch, the loss in the post + fill launch, one launch for both list sorts, the clip norm out of the last
column sum, the bias vectors' Adam at the end of the encoder's) against the step of single launches (sae_fold = 0).  Alternating legs on
one box; step time + loss (the two forms agree bit for bit: tests/test_native_sae_gpu.py)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vit_prisma_amd import _native as N
from vit_prisma_amd.sae.bench_leg import sae_bench_leg

from vit_prisma_amd.sae.native_sae import NativeSAE

_step = NativeSAE.step
FOLD = [1]


def step_ab(self, *a, **kw):
    # the leg of single launches takes its clip norm from pv_sae_grad_sqnorm_step, as the step did before the folds
    if not FOLD[0]:
        kw["fused_sqnorm"] = False
    return _step(self, *a, **kw)


NativeSAE.step = step_ab
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
rows = []
MODES = (0, 1)
for rep in range(int(os.environ.get("REPS", "3"))):
    for side in MODES:
        N.set_tuning("reset")
        N.set_tuning("sae_fold", side)
        FOLD[0] = side
        r = sae_bench_leg(dev, dist=None, steps=int(os.environ.get("STEPS", "40")), warmup=5)
        rows.append({"sae_fold": side, "ms_per_step": r["ms_per_step"], "final_loss": r.get("final_loss"), "l0": r.get("l0")})
        print(rows[-1], flush=True)
N.set_tuning("reset")
for side in MODES:
    v = sorted(x["ms_per_step"] for x in rows if x["sae_fold"] == side)
    print(f"sae_fold={side}: median {v[len(v) // 2]:.4f} ms  (min {v[0]:.4f}, max {v[-1]:.4f})")
print(json.dumps(rows))
