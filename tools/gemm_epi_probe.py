"""How long does ONE tile's store epilogue take when few CUs run it (no fabric contention) against a full launch?  Phase trace
(pv_debug_gemm_trace_*) of standalone pv_gemm_epilogue launches: K loop / epilogue per tile, by epilogue kind and tile count."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vit_prisma_amd import _native as N
L = N.lib()
L.pv_debug_gemm_trace_arm.argtypes = [ctypes.c_int32]
L.pv_debug_gemm_trace_read.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p]
dev = torch.device("cuda:0"); st = torch.cuda.current_stream().cuda_stream
EPI = {"bias": 0, "resid": 2, "act": 3}
persist = int(os.environ.get("PV_PERSIST", "0"))
N.set_tuning("gemm_persist", persist)
if os.environ.get("PV_TILE"): N.set_tuning("gemm_tile", int(os.environ["PV_TILE"]))
print("gemm_persist", persist)
for epi in ("bias", "act", "resid"):
    for M, Nn, K in ((320 * 8, 256, 768), (320 * 32, 256, 768), (320 * 80, 768, 768), (320 * 80, 3072, 768), (320 * 8, 256 * 4, 768)):
        A = torch.randn(M, K, device=dev).bfloat16(); B = (torch.randn(Nn, K, device=dev) * 0.05).bfloat16()
        bias = torch.randn(Nn, device=dev).bfloat16(); res = torch.randn(M, Nn, device=dev).bfloat16()
        o0 = torch.zeros(M, Nn, device=dev, dtype=torch.bfloat16); o1 = torch.zeros_like(o0)
        def go():
            N.check(L.pv_gemm_epilogue(1, EPI[epi], 0, A.data_ptr(), K, B.data_ptr(), K, bias.data_ptr(), res.data_ptr(), Nn, o0.data_ptr(), o1.data_ptr(), Nn, M, Nn, K, st), "gemm")
        for _ in range(3): go()
        torch.cuda.synchronize()
        assert L.pv_debug_gemm_trace_arm(0) == 0
        go()
        buf = np.zeros((8192, 4), np.uint64); info = np.zeros(6, np.int32)
        assert L.pv_debug_gemm_trace_read(buf.ctypes.data, 8192, info.ctypes.data) == 0
        n = int(info[4]); t = buf[:n].astype(np.int64); ok = t[:, 2] > 0
        t0 = t[ok, 0].min()
        stt, le, en = (t[ok, 0] - t0) / 100.0, (t[ok, 1] - t0) / 100.0, (t[ok, 2] - t0) / 100.0
        loop, ep = le - stt, en - le
        print(f"{epi:5s} {M}x{Nn}x{K} tiles={n:4d} kernel=v{info[5]} span {en.max():6.1f} us | loop p50 {np.percentile(loop,50):6.2f} | epilogue p10 {np.percentile(ep,10):6.2f} p50 {np.percentile(ep,50):6.2f} p90 {np.percentile(ep,90):6.2f}", flush=True)
N.set_tuning("reset")
