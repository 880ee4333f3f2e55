"""GEMM micro-benchmark on the B/32 bs=512 shapes with one tile kernel forced (PV_TILE = 5: eight waves, 9: four waves)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vit_prisma_amd import _native as N
L = N.lib()
N.set_tuning("gemm_tile", int(os.environ.get("PV_TILE", "5")))
dev = torch.device("cuda:0")
shapes = [("qkv", 25600, 2304, 768), ("oproj", 25600, 768, 768), ("mlp1", 25600, 3072, 768), ("mlp2", 25600, 768, 3072), ("sq8192", 8192, 8192, 8192)]
reps = int(os.environ.get("REPS", "30"))
for name, M, Nn, K in shapes:
    A = torch.randn(M, K, device=dev).bfloat16(); B = torch.randn(Nn, K, device=dev).bfloat16()
    bias = torch.randn(Nn, device=dev).bfloat16(); C = torch.empty(M, Nn, device=dev, dtype=torch.bfloat16)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        N.check(L.pv_gemm_bias(1, A.data_ptr(), K, B.data_ptr(), K, bias.data_ptr(), C.data_ptr(), Nn, M, Nn, K, st), "gemm")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        L.pv_gemm_bias(1, A.data_ptr(), K, B.data_ptr(), K, bias.data_ptr(), C.data_ptr(), Nn, M, Nn, K, st)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    ref = (A[:256].float() @ B.float().T + bias.float())
    err = float((C[:256].float() - ref).abs().max() / ref.abs().max())
    print(f"tile {os.environ.get('PV_TILE')} {name:8s} {M}x{Nn}x{K}: {us:8.1f} us  {2.0*M*Nn*K/us/1e6:7.1f} TFLOP/s  err {err:.2e}", flush=True)
