"""Is the bs=512 all-hooks forward's inter-kernel gap (~4.6 us x 102 launches = 7 % of the step) something a HIP graph removes?
Captures ONE native run_with_cache into a torch.cuda.CUDAGraph (hipGraph underneath) and times replay against eager launches."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vit_prisma_amd import HookedViT, HookedViTConfig
from vit_prisma_amd.synth import ARCHS, synth_vit_state

dev = torch.device("cuda:0")
arch = ARCHS["clip-vit-b32"]
model = HookedViT(HookedViTConfig(**arch, dtype=torch.bfloat16, device="cuda"))
model.load_state_dict({k: torch.from_numpy(v) for k, v in synth_vit_state(arch, 0).items()}, strict=True)
model = model.to(torch.bfloat16).to(dev).eval().use_native(True)
BS = int(os.environ.get("BS", "512"))
images = torch.randn(BS, 3, 224, 224, device=dev).bfloat16()


def timed(fn, n=10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


with torch.no_grad():
    for _ in range(3):
        out, cache = model.run_with_cache(images)
    ref = {k: v.clone() for k, v in cache.items()}
    eager = [timed(lambda: model.run_with_cache(images)) for _ in range(3)]
    print("eager ms/step", [round(t, 3) for t in eager])
    try:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                model.run_with_cache(images)
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out_g, cache_g = model.run_with_cache(images)
        g.replay()
        torch.cuda.synchronize()
        same = all(torch.equal(cache_g[k], ref[k]) for k in ref)
        graph = [timed(g.replay) for _ in range(3)]
        print("graph ms/step", [round(t, 3) for t in graph], "identical cache:", same)
    except Exception as e:
        print("graph capture failed:", type(e).__name__, str(e)[:300])
