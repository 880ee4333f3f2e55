"""The bs=512 B/32 all-hooks forward with and without launch-graph replay (NativeViT.use_graphs), alternating rounds, + bit-identity of
every cache entry; also bs=32 (a store-sized harvest forward) where launch gaps weigh more."""
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


def timed(fn, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


with torch.no_grad():
    for bs, n in ((512, 10), (32, 50)):
        images = torch.randn(bs, 3, 224, 224, device=dev).bfloat16()
        for _ in range(3):
            out, cache = model.run_with_cache(images)
        nv = model._native
        nv.use_graphs = False
        out, cache = model.run_with_cache(images)
        ref = {k: v.clone() for k, v in cache.items()}
        del out, cache
        res = {False: [], True: []}
        for rnd in range(3):
            for g in (False, True):
                nv.use_graphs = g
                for _ in range(3):
                    model.run_with_cache(images)
                res[g].append(timed(lambda: model.run_with_cache(images), n))
        nv.use_graphs = True
        out, cache = model.run_with_cache(images)
        same = all(torch.equal(cache[k], ref[k]) for k in ref)
        print(f"bs={bs}: eager ms/forward {[round(t, 3) for t in res[False]]} | graph replay {[round(t, 3) for t in res[True]]} | "
              f"{bs / min(res[False]) :.1f} -> {bs / min(res[True]):.1f} k images/s | all {len(ref)} cache entries bit-identical: {same} | "
              f"replays so far {nv.n_graph_replays}", flush=True)
        del out, cache, ref
