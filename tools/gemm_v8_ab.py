"""A/B of the two-workgroups-per-CU GEMM (tuning gemm_tile=8, gemm_prio 0..3) against the one-workgroup kernel:
(A) bias epilogue on the B/32 shapes + ragged ones: bit-equality and time per launch; (B) the whole bs=512 all-hooks forward:
every cache entry bit-identical, time per step; (C) phase trace of one O-projection / MLP-1 launch with the workgroup slot ids
(HW_ID) of the two workgroups of each CU."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from vit_prisma_amd import HookedViT, HookedViTConfig, _native as N
from vit_prisma_amd.synth import ARCHS, synth_vit_state

L = N.lib()
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
reps = int(os.environ.get("REPS", "20"))
variants = [(5, 1), (8, 0), (8, 1), (8, 2), (8, 3)]
parts = os.environ.get("PARTS", "ABC")

def timed(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n

if "A" in parts:
    shapes = [("qkv", 25600, 2304, 768), ("oproj", 25600, 768, 768), ("mlp1", 25600, 3072, 768), ("mlp2", 25600, 768, 3072),
              ("ragged", 25600 - 37, 1000, 1024), ("small", 777, 264, 96), ("sq4096", 4096, 4096, 4096)]
    torch.manual_seed(0)
    for name, M, Nn, K in shapes:
        A = torch.randn(M, K, device=dev).bfloat16(); B = (torch.randn(Nn, K, device=dev) * 0.05).bfloat16()
        bias = torch.randn(Nn, device=dev).bfloat16()
        ref = None; row = []
        for tile, prio in variants:
            N.set_tuning("gemm_tile", tile); N.set_tuning("gemm_prio", prio)
            C = torch.zeros(M, Nn, device=dev, dtype=torch.bfloat16)
            call = lambda: N.check(L.pv_gemm_bias(1, A.data_ptr(), K, B.data_ptr(), K, bias.data_ptr(), C.data_ptr(), Nn, M, Nn, K, st), "gemm")
            for _ in range(3):
                call()
            torch.cuda.synchronize()
            us = timed(call, reps)
            if ref is None:
                ref = C.clone(); same = "ref"
            else:
                same = "same" if torch.equal(ref.view(torch.int16), C.view(torch.int16)) else f"DIFFERS {float((ref.float() - C.float()).abs().max()):.3e}"
            row.append(f"t{tile}p{prio}: {us:7.1f} us {2.0 * M * Nn * K / us / 1e6:5.0f} TF [{same}]")
        print(f"{name:7s} {M}x{Nn}x{K}: " + " | ".join(row), flush=True)
    N.set_tuning("reset")

if "B" in parts or "C" in parts:
    arch = ARCHS["clip-vit-b32"]
    model = HookedViT(HookedViTConfig(**arch, dtype=torch.bfloat16, device="cuda"))
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth_vit_state(arch, 0).items()}, strict=True)
    model = model.to(torch.bfloat16).to(dev).eval().use_native(True)
    images = torch.randn(512, 3, 224, 224, device=dev, generator=torch.Generator(device=dev).manual_seed(1234)).bfloat16()

if "B" in parts:
    with torch.no_grad():
        ref = None
        for tile, prio in [(-1, 1)] + variants[1:] + [(-1, 1)]:
            N.set_tuning("gemm_tile", tile); N.set_tuning("gemm_prio", prio)
            for _ in range(3):
                out, cache = model.run_with_cache(images)
            snap = {k: v.clone() for k, v in cache.items()}
            del out, cache
            torch.cuda.synchronize()
            def step():
                o, c = model.run_with_cache(images)
                del o, c
            ms = timed(step, 15) / 1e3
            if ref is None:
                ref = snap; same = "ref"
            else:
                bad = [k for k in ref if not torch.equal(ref[k].view(torch.uint8) if ref[k].is_contiguous() else ref[k].contiguous().view(torch.uint8),
                                                         snap[k].view(torch.uint8) if snap[k].is_contiguous() else snap[k].contiguous().view(torch.uint8))]
                same = "all 214 entries bit-identical" if not bad else f"{len(bad)} entries DIFFER, e.g. {bad[:3]}"
            print(f"forward tile={tile} prio={prio}: {ms:7.3f} ms/step = {512 / ms * 1e3:8.0f} images/s [{same}]", flush=True)
            del snap
        N.set_tuning("reset")

if "C" in parts:
    L.pv_debug_gemm_trace_arm.argtypes = [ctypes.c_int32]
    L.pv_debug_gemm_trace_read.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p]
    names = ["qkv", "oproj", "mlp1", "mlp2"]
    with torch.no_grad():
        for tile, prio in [(-1, 1), (8, 0), (8, 1)]:
            N.set_tuning("gemm_tile", tile); N.set_tuning("gemm_prio", prio)
            for _ in range(2):
                model.run_with_cache(images)
            torch.cuda.synchronize()
            for j in (0, 1, 2, 3):
                assert L.pv_debug_gemm_trace_arm(5 * 4 + j) == 0
                model.run_with_cache(images)
                buf = np.zeros((8192, 4), np.uint64); info = np.zeros(6, np.int32)
                assert L.pv_debug_gemm_trace_read(buf.ctypes.data, 8192, info.ctypes.data) == 0
                n = int(info[4]); t = buf[:n].astype(np.int64)
                ok = t[:, 2] > 0
                t0 = t[ok, 0].min()
                s0, le, en = (t[ok, 0] - t0) / 100.0, (t[ok, 1] - t0) / 100.0, (t[ok, 2] - t0) / 100.0
                hw = (buf[:n, 3][ok] & np.uint64(0xffffffff)).astype(np.int64); xcc = (buf[:n, 3][ok] >> np.uint64(32)).astype(np.int64) & 0xf
                tg = (hw >> 16) & 0xf; wv = hw & 0xf; cu = (xcc << 12) | (hw & 0xff00)
                hi = s0 < 1.0                                                   # first-round workgroups
                print(f"tile={tile} prio={prio} {names[j]:6s} kernel=v{info[5]} wgs={n} span={en.max():6.1f} us | loop mean {np.mean(le - s0):6.2f}"
                      f" epi mean {np.mean(en - le):6.2f} | first round: loop-end p10/p50/p90 {np.percentile(le[hi], 10):5.1f}/{np.percentile(le[hi], 50):5.1f}/{np.percentile(le[hi], 90):5.1f}"
                      f" end p50/p90 {np.percentile(en[hi], 50):5.1f}/{np.percentile(en[hi], 90):5.1f}", flush=True)
                if tile == 8 and j == 1:
                    print("    TG_ID histogram:", np.bincount(tg, minlength=4)[:8].tolist(), " WAVE_ID (wave 0) histogram:", np.bincount(wv, minlength=4)[:10].tolist(),
                          " distinct CUs:", len(np.unique(cu)))
                    for bit, nm in ((tg & 1, "TG_ID&1"), (wv & 1, "WAVE_ID&1")):
                        a, b = le[hi & (bit == 0)], le[hi & (bit == 1)]
                        if len(a) and len(b):
                            print(f"    loop end by {nm}: bit0 n={len(a)} mean {a.mean():5.1f} us | bit1 n={len(b)} mean {b.mean():5.1f} us")
        N.set_tuning("reset")
