"""Per-workgroup phase trace of one GEMM launch INSIDE the bs=512 B/32 all-hooks forward
(pv_debug_gemm_trace_*): how long the K loop and the store epilogue of each tile really take, and how much
of the launch has workgroups in the loop / in the epilogue at the same time."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from vit_prisma_amd import HookedViT, HookedViTConfig, _native as N
from vit_prisma_amd.synth import ARCHS, synth_vit_state

dev = torch.device("cuda:0")
arch = ARCHS["clip-vit-b32"]
model = HookedViT(HookedViTConfig(**arch, dtype=torch.bfloat16, device="cuda"))
model.load_state_dict({k: torch.from_numpy(v) for k, v in synth_vit_state(arch, 0).items()}, strict=True)
model = model.to(torch.bfloat16).to(dev).eval().use_native(True)
BS = int(os.environ.get('BS', '512'))
images = torch.randn(BS, 3, 224, 224, device=dev).bfloat16()
L = N.lib()
if os.environ.get('PV_PERSIST') is not None:
    N.set_tuning('gemm_persist', int(os.environ['PV_PERSIST']))
    print('gemm_persist =', os.environ['PV_PERSIST'])
L.pv_debug_gemm_trace_arm.argtypes = [ctypes.c_int32]
L.pv_debug_gemm_trace_read.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p]
with torch.no_grad():
    for _ in range(3):
        model.run_with_cache(images)
    torch.cuda.synchronize()
    names = ["qkv", "oproj", "mlp1", "mlp2"]
    for j in [int(x) for x in os.environ.get('WHICH', '0,1,2,3').split(',')]:
        assert L.pv_debug_gemm_trace_arm(5 * 4 + j) == 0
        model.run_with_cache(images)
        buf = np.zeros((8192, 4), np.uint64); info = np.zeros(6, np.int32)
        assert L.pv_debug_gemm_trace_read(buf.ctypes.data, 8192, info.ctypes.data) == 0
        n = int(info[4]); t = buf[:n].astype(np.int64)
        ok = t[:, 2] > 0
        t0 = t[ok, 0].min()
        st, le, en = (t[ok, 0] - t0) / 100.0, (t[ok, 1] - t0) / 100.0, (t[ok, 2] - t0) / 100.0     # us
        span = en.max()
        loop, epi = le - st, en - le
        # time-resolved concurrency (0.1 us bins)
        grid = np.arange(0, span, 0.1)
        in_loop = ((st[None, :] <= grid[:, None]) & (grid[:, None] < le[None, :])).sum(1)
        in_epi = ((le[None, :] <= grid[:, None]) & (grid[:, None] < en[None, :])).sum(1)
        print(f"{names[j]:6s} M,N,K={info[0]},{info[1]},{info[2]} epi={info[3]} kernel=v{info[5]} wgs={n} traced={int(ok.sum())} span={span:7.1f} us")
        print(f"    loop us: mean {loop.mean():6.2f} p10 {np.percentile(loop,10):6.2f} p50 {np.percentile(loop,50):6.2f} p90 {np.percentile(loop,90):6.2f}"
              f" | epilogue us: mean {epi.mean():6.2f} p10 {np.percentile(epi,10):6.2f} p50 {np.percentile(epi,50):6.2f} p90 {np.percentile(epi,90):6.2f}")
        print(f"    avg resident WGs: in loop {in_loop.mean():6.1f}, in epilogue {in_epi.mean():6.1f}; "
              f"time with >=1/2 of max residency in epilogue: {(in_epi > 0.5 * (in_loop + in_epi).max()).mean():.2f}; first-start spread {np.percentile(st,90):.1f} us (p90)")
        q = np.linspace(0, span, 9)
        print("    timeline (WGs in loop/in epilogue at 8 points):", " ".join(f"{in_loop[min(int(x/0.1), len(grid)-1)]}/{in_epi[min(int(x/0.1), len(grid)-1)]}" for x in q[:-1]))
