"""Host time of one all-hooks run_with_cache call at a small batch (the forward is CPU-bound there): cProfile of 200 calls."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vit_prisma_amd import HookedViT, HookedViTConfig
from vit_prisma_amd.synth import ARCHS, synth_vit_state

dev = torch.device("cuda:0")
arch = ARCHS["clip-vit-b32"]
model = HookedViT(HookedViTConfig(**arch, dtype=torch.bfloat16, device="cuda"))
model.load_state_dict({k: torch.from_numpy(v) for k, v in synth_vit_state(arch, 0).items()}, strict=True)
model = model.to(torch.bfloat16).to(dev).eval().use_native(True)
bs = int(os.environ.get("BS", "32"))
images = torch.randn(bs, 3, 224, 224, device=dev).bfloat16()
with torch.no_grad():
    for _ in range(5):
        model.run_with_cache(images)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        model.run_with_cache(images)
    t_host = (time.perf_counter() - t0) / 200 * 1e3
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) / 200 * 1e3
    print(f"bs={bs}: host {t_host:.3f} ms per call issued, {t_all:.3f} ms per call completed")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(200):
        model.run_with_cache(images)
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("cumulative").print_stats(28)
