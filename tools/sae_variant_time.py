"""Step time of one coder variant of sae_variants_leg (VARIANT=gated_relu_l0_64 | gated_relu | transcoder_topk_skip), as JSON."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vit_prisma_amd.sae.bench_leg import sae_variants_leg

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
print(json.dumps(sae_variants_leg(dev, steps=int(os.environ.get("STEPS", "20")), warmup=int(os.environ.get("WARMUP", "5")),
                                  only=os.environ.get("VARIANT", "gated_relu_l0_64"))))
