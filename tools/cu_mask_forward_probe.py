"""Probe: does the all-hooks B/32 forward gain from running as TWO half-batch chains side by side, each on its own
share of the chip, so that one half's HBM / latency-bound kernels (LayerNorm, attention, launch gaps) run beside the other half's
power-capped GEMMs instead of behind them?

Streams come from hipExtStreamCreateWithCUMask (bit b of the 256-bit mask = CU b / 8 of XCC b % 8 -- tools/probes/cu_mask_probe.hip --
so a mask of n contiguous bits is n / 8 CUs of EVERY XCD and the kernels' blockIdx % 8 -> XCD order survives); the persistent GEMM is told
how many CUs its stream has (tuning key gemm_cus).  Two model replicas with the same weights, each with its own plan, workspace and
arena; total images per pass = BATCH in every variant.  Shader clock and socket power are sampled from rocm-smi while each variant loops.

    python tools/cu_mask_forward_probe.py            # B/32, bs=512
    ARCH=clip-vit-l14-336 BATCH=128 python tools/cu_mask_forward_probe.py
"""
import ctypes as C
import os
import re
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vit_prisma_amd import HookedViT, HookedViTConfig, _native as N
from vit_prisma_amd.synth import ARCHS, synth_vit_state

dev = torch.device("cuda", 0)
B = int(os.environ.get("BATCH", "512"))
steps = int(os.environ.get("STEPS", "30"))
arch_name = os.environ.get("ARCH", "clip-vit-b32")
arch = ARCHS[arch_name]
pattern_only = arch_name != "clip-vit-b32"
sd = {k: torch.from_numpy(v) for k, v in synth_vit_state(arch, 0).items()}
hip = C.CDLL("libamdhip64.so")
hip.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]


def masked_stream(lo: int, hi: int) -> torch.cuda.Stream:
    words = (C.c_uint32 * 8)()
    for b in range(lo, hi):
        words[b >> 5] |= 1 << (b & 31)
    h = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(h), 8, words)
    assert rc == 0 and h.value, rc
    return torch.cuda.ExternalStream(h.value, device=dev)


def make():
    m = HookedViT(HookedViTConfig(**arch, dtype=torch.bfloat16, device="cuda"))
    m.load_state_dict(sd, strict=True)
    return m.to(torch.bfloat16).to(dev).eval().use_native(True)


samples = []
stop = False


def sampler():
    while not stop:
        try:
            o = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=5).stdout
            sclk = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", o)
            pw = re.search(r"Power \(W\): ([\d.]+)", o)
            samples.append((time.time(), int(sclk.group(1)) if sclk else -1, float(pw.group(1)) if pw else -1.0))
        except Exception:
            samples.append((time.time(), -2, -2.0))


g = torch.Generator(device=dev).manual_seed(1234)
images = torch.randn(B, 3, arch["image_size"], arch["image_size"], device=dev, generator=g).to(torch.bfloat16)
kw = dict(names_filter=lambda n: n.endswith("attn.hook_pattern")) if pattern_only else {}


def timed(fn, label, min_seconds=2.5):
    with torch.no_grad():
        for _ in range(4):
            fn()
        torch.cuda.synchronize()
        n0 = len(samples)
        t0 = time.perf_counter()
        n = 0
        while n < steps or time.perf_counter() - t0 < min_seconds:
            fn()
            n += 1
            if n % 8 == 0:
                torch.cuda.synchronize()           # (keeps the host from queueing seconds of work ahead)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
    ss = [s for s in samples[n0:] if s[1] > 0]
    clk, pw = sorted(s[1] for s in ss), sorted(s[2] for s in ss)
    print(f"{label:58s} {dt * 1e3:7.3f} ms/pass {B / dt:9.1f} images/s | sclk median {clk[len(clk) // 2] if clk else -1} MHz, "
          f"power median {pw[len(pw) // 2] if pw else -1:.0f} W ({len(clk)} samples)", flush=True)
    return dt


th = threading.Thread(target=sampler)
th.start()
try:
    one = make()
    two = make()
    base = timed(lambda: one.run_with_cache(images, **kw), f"one stream, bs={B} (the shipped path)")
    chunks = list(images.chunk(2))

    def pair(streams, delay_cycles=0):
        def run():
            cur = torch.cuda.current_stream(dev)
            for s in streams:
                s.wait_stream(cur)
            keep = []
            for i, (m, s, x) in enumerate(zip((one, two), streams, chunks)):
                with torch.cuda.stream(s):
                    if i == 1 and delay_cycles:
                        torch.cuda._sleep(delay_cycles)
                    keep.append(m.run_with_cache(x, **kw))
            for s in streams:
                cur.wait_stream(s)
            return keep
        return run

    timed(lambda: [m.run_with_cache(x, **kw) for m, x in zip((one, two), chunks)], f"one stream, 2 x bs={B // 2} back to back")
    plain = [torch.cuda.Stream(device=dev) for _ in range(2)]
    timed(pair(plain), f"two plain streams, 2 x bs={B // 2}")
    for cus in (128, 96):
        N.set_tuning("gemm_cus", cus)
        timed(pair(plain), f"two plain streams, GEMM grids of {cus} CUs")
        N.set_tuning("reset")
    # (lo_a, hi_a, lo_b, hi_b, GEMM grid): disjoint halves; then shares that overlap in the middle -- a chain's GEMMs take 128 CUs, its
    # LayerNorm / attention kernels may spread over more
    for lo_a, hi_a, lo_b, hi_b, cus in ((0, 128, 128, 256, 128), (0, 160, 96, 256, 128), (0, 192, 64, 256, 128), (0, 144, 112, 256, 112)):
        sa, sb = masked_stream(lo_a, hi_a), masked_stream(lo_b, hi_b)
        N.set_tuning("gemm_cus", cus)
        label = f"masked streams CUs [{lo_a},{hi_a}) | [{lo_b},{hi_b}), GEMM grids {cus}"
        timed(pair([sa, sb]), label)
        if hi_a == 128 and not pattern_only:
            for d in (200_000, 600_000, 1_500_000):
                timed(pair([sa, sb], d), f"  ... second chain delayed by {d} clocks")
        N.set_tuning("reset")
finally:
    stop = True
    th.join()
