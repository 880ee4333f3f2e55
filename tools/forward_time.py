"""ms per step of the bs = 512 B/32 all-hooks forward (the bench's main line) with the library PV_NATIVE_LIB names: A/B of kernel builds in
alternating processes on one box.   python tools/forward_time.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vit_prisma_amd import HookedViT, HookedViTConfig
from vit_prisma_amd.synth import ARCHS, synth_vit_state
dev = torch.device("cuda:0")
arch = ARCHS["clip-vit-b32"]
model = HookedViT(HookedViTConfig(**arch, dtype=torch.bfloat16, device="cuda"))
model.load_state_dict({k: torch.from_numpy(v) for k, v in synth_vit_state(arch, 0).items()}, strict=True)
model = model.to(torch.bfloat16).to(dev).eval().use_native(True)
images = torch.randn(512, 3, 224, 224, device=dev).bfloat16()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
with torch.no_grad():
    for _ in range(5):
        model.run_with_cache(images)
    best = []
    for r in range(3):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            out, cache = model.run_with_cache(images)
        e1.record(); torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) / reps)
print(f"{os.environ.get('PV_NATIVE_LIB', 'in-tree')}: ms/step {min(best):.3f} (runs {', '.join(f'{b:.3f}' for b in best)}) = {512 / min(best) * 1e3:.0f} images/s")
