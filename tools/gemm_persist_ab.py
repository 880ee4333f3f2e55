"""Persistent (v8) against one-tile-per-workgroup (v7) form of the bf16 GEMM: bit-equality of every output on the forward's
shapes (+ ragged ones) for the bias / activation / residual epilogues, time per launch (interleaved rounds), and the bs=512
B/32 forward under either.
    python tools/gemm_persist_ab.py [quick]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vit_prisma_amd import _native as N

L = N.lib()
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
reps = int(os.environ.get("REPS", "20"))
rounds = int(os.environ.get("ROUNDS", "3"))
EPI = {"bias": 0, "resid": 2, "act": 3}
shapes = [("qkv", "bias", 25600, 2304, 768), ("oproj", "resid", 25600, 768, 768), ("mlp1", "act", 25600, 3072, 768),
          ("mlp2", "resid", 25600, 768, 3072), ("ragged", "act", 25600 - 37, 1000, 1024), ("raggedr", "resid", 8000 + 13, 1024, 256),
          ("l14o", "resid", 128 * 577, 1024, 1024), ("l14m1", "act", 128 * 577, 4096, 1024)]
if len(sys.argv) > 1 and sys.argv[1] == "quick":
    shapes = shapes[:5]
torch.manual_seed(0)


def run(epi, A, B, bias, res, o0, o1, M, Nn, K):
    N.check(L.pv_gemm_epilogue(1, EPI[epi], 0, A.data_ptr(), K, B.data_ptr(), K, bias.data_ptr(), res.data_ptr() if res is not None else None, Nn,
                               o0.data_ptr(), o1.data_ptr() if o1 is not None else None, Nn, M, Nn, K, st), "gemm")


variants = [("v7", 0, 0), ("v8", 1, 0)]            # (name, gemm_persist, gemm_stagger)
for name, epi, M, Nn, K in shapes:
    A = torch.randn(M, K, device=dev).bfloat16(); B = (torch.randn(Nn, K, device=dev) * 0.05).bfloat16()
    bias = torch.randn(Nn, device=dev).bfloat16()
    res = torch.randn(M, Nn, device=dev).bfloat16() if epi == "resid" else None
    outs = {}
    for vn, persist, stag in variants[:2]:
        N.set_tuning("gemm_persist", persist); N.set_tuning("gemm_stagger", stag)
        o0 = torch.zeros(M, Nn, device=dev, dtype=torch.bfloat16); o1 = torch.zeros_like(o0) if epi != "bias" else None
        run(epi, A, B, bias, res, o0, o1, M, Nn, K)
        torch.cuda.synchronize()
        outs[vn] = (o0, o1)
    same = all(torch.equal(a.view(torch.int16), b.view(torch.int16)) for a, b in zip(outs["v7"], outs["v8"]) if a is not None)
    # spot check against torch on the first rows
    ref = A[:256].float() @ B.float().T + bias.float()
    chk = outs["v8"][0][:256].float()
    err = float((chk - ref).abs().max() / ref.abs().max())
    times = {vn: [] for vn, _, _ in variants}
    o0 = torch.zeros(M, Nn, device=dev, dtype=torch.bfloat16); o1 = torch.zeros_like(o0) if epi != "bias" else None
    for r in range(rounds):
        for vn, persist, stag in variants:
            N.set_tuning("gemm_persist", persist); N.set_tuning("gemm_stagger", stag)
            run(epi, A, B, bias, res, o0, o1, M, Nn, K)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                run(epi, A, B, bias, res, o0, o1, M, Nn, K)
            e1.record(); torch.cuda.synchronize()
            times[vn].append(e0.elapsed_time(e1) * 1e3 / reps)
    row = " | ".join(f"{vn} {min(t):7.1f}/{sorted(t)[len(t)//2]:7.1f} us" for vn, t in times.items())
    print(f"{name:8s} {epi:5s} {M}x{Nn}x{K}: {'bit-identical' if same else 'DIFFERS'} err {err:.1e} | min/med: {row}", flush=True)
N.set_tuning("reset")

# ---- the forward
from vit_prisma_amd import HookedViT, HookedViTConfig
from vit_prisma_amd.synth import ARCHS, synth_vit_state
arch = ARCHS["clip-vit-b32"]
model = HookedViT(HookedViTConfig(**arch, dtype=torch.bfloat16, device="cuda"))
model.load_state_dict({k: torch.from_numpy(v) for k, v in synth_vit_state(arch, 0).items()}, strict=True)
model = model.to(torch.bfloat16).to(dev).eval().use_native(True)
images = torch.randn(512, 3, 224, 224, device=dev).bfloat16()
ref_cache = None
with torch.no_grad():
    for tag, persist, stag in [("auto", -1, 0), ("off", 0, 0), ("all", 1, 0), ("auto", -1, 0), ("off", 0, 0), ("all", 1, 0)]:
        N.set_tuning("reset"); N.set_tuning("gemm_persist", persist); N.set_tuning("gemm_stagger", stag)
        for _ in range(3):
            out, cache = model.run_with_cache(images)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            out, cache = model.run_with_cache(images)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        note = ""
        if ref_cache is None:
            ref_cache = {k: v.clone() for k, v in cache.items()}
        else:
            bad = [k for k, v in cache.items() if not torch.equal(v, ref_cache[k])]
            note = "all entries bit-identical" if not bad else f"{len(bad)} entries DIFFER e.g. {bad[:3]}"
        print(f"forward persist={tag:8s}: {ms:7.3f} ms/step = {512 / ms * 1e3:8.0f} images/s  {note}", flush=True)
N.set_tuning("reset")
