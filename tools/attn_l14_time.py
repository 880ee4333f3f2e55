"""Attention time per layer of the L/14@336 bs=128 pattern-only forward (HIP events around the attention launches) + step time."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vit_prisma_amd import HookedViT, HookedViTConfig, _native as N
from vit_prisma_amd.synth import ARCHS, synth_vit_state
dev = torch.device("cuda:0")
arch = ARCHS["clip-vit-l14-336"]
model = HookedViT(HookedViTConfig(**arch, dtype=torch.bfloat16, device="cuda"))
model.load_state_dict({k: torch.from_numpy(v) for k, v in synth_vit_state(arch, 0).items()}, strict=True)
model = model.to(torch.bfloat16).to(dev).eval().use_native(True)
images = torch.randn(128, 3, 336, 336, device=dev, generator=torch.Generator(device=dev).manual_seed(4321)).to(torch.bfloat16)
out = {}
with torch.no_grad():
    for name, keep in (("pattern_only", lambda n: n.endswith("attn.hook_pattern")), ("z_only", lambda n: n.endswith("attn.hook_z"))):
        for _ in range(2):
            o, c = model.run_with_cache(images, names_filter=keep); del o, c
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(4):
            o, c = model.run_with_cache(images, names_filter=keep); del o, c
        torch.cuda.synchronize()
        step = (time.perf_counter() - t0) / 4 * 1e3
        N.prof_reset(); N.prof_enable(True, kinds=("attention",))
        for _ in range(2):
            o, c = model.run_with_cache(images, names_filter=keep); del o, c
        torch.cuda.synchronize(); N.prof_enable(False)
        r = N.prof_read("attention")
        out[name] = {"ms_per_step": round(step, 2), "attention_us_per_layer": round(r["ms"] / r["launches"] * 1e3, 1), "launches": r["launches"]}
print(json.dumps(out))
