"""Diagnosis runs for the T = 577 attention kernels: L/14@336-shaped model with few layers, bs = 128.
env: MODE = pattern | none | z ; LAYERS ; ITERS"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vit_prisma_amd import HookedViT, HookedViTConfig
from vit_prisma_amd.synth import ARCHS

mode = os.environ.get("MODE", "pattern")
arch = dict(ARCHS["clip-vit-l14-336"]); arch["n_layers"] = int(os.environ.get("LAYERS", "4"))
model = HookedViT(HookedViTConfig(**arch, dtype=torch.bfloat16, device="cuda")).to(torch.bfloat16).cuda().eval().use_native(True)
x = torch.randn(128, 3, 336, 336, device="cuda").bfloat16()
with torch.no_grad():
    for _ in range(int(os.environ.get("ITERS", "3"))):
        if mode == "pattern":
            model.run_with_cache(x, names_filter=lambda n: n.endswith("attn.hook_pattern"))
        elif mode == "z":
            model.run_with_cache(x, names_filter=lambda n: n.endswith("attn.hook_z"))
        else:
            model(x)
        torch.cuda.synchronize()
assert model.last_run_native, model.native_fallback_reason
print("ok", mode)
