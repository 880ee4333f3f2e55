"""Probe: does the all-hooks B/32 forward gain from running the batch as TWO independent half-batches on two HIP streams?
(Independent kernel chains interleave on the GPU: one half's store-heavy epilogues / LayerNorms / attention against the other half's
K loops, the partial last round of one GEMM filled by the other half's workgroups.)  Two model replicas with the same weights, each
with its own plan, workspace and arena; total images per pass = BATCH in every variant."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vit_prisma_amd import HookedViT, HookedViTConfig
from vit_prisma_amd.synth import ARCHS, synth_vit_state

dev = torch.device("cuda", 0)
B = int(os.environ.get("BATCH", "512"))
steps = int(os.environ.get("STEPS", "20"))
arch = ARCHS["clip-vit-b32"]
sd = {k: torch.from_numpy(v) for k, v in synth_vit_state(arch, 0).items()}


def make():
    m = HookedViT(HookedViTConfig(**arch, dtype=torch.bfloat16, device="cuda"))
    m.load_state_dict(sd, strict=True)
    return m.to(torch.bfloat16).to(dev).eval().use_native(True)


g = torch.Generator(device=dev).manual_seed(1234)
images = torch.randn(B, 3, 224, 224, device=dev, generator=g).to(torch.bfloat16)


def timed(fn, label):
    with torch.no_grad():
        for _ in range(4):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
    print(f"{label:44s} {dt * 1e3:7.3f} ms/pass  {B / dt:9.1f} images/s", flush=True)


one = make()
timed(lambda: one.run_with_cache(images), f"one stream, bs={B}")
for parts in (2, 4):
    models = [one] + [make() for _ in range(parts - 1)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(parts)]
    chunks = list(images.chunk(parts))

    def run_split():
        cur = torch.cuda.current_stream(dev)
        for s in streams:
            s.wait_stream(cur)
        keep = []
        for m, s, x in zip(models, streams, chunks):
            with torch.cuda.stream(s):
                keep.append(m.run_with_cache(x))
        for s in streams:
            cur.wait_stream(s)
        return keep

    def run_serial():
        return [m.run_with_cache(x) for m, x in zip(models, chunks)]

    timed(run_serial, f"one stream, {parts} x bs={B // parts} back to back")
    timed(run_split, f"{parts} streams, {parts} x bs={B // parts} concurrently")
