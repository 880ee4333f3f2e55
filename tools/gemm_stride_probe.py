"""Does the operand row stride bound the K loop?  pv_gemm_bias on the B/32 shapes with padded lda / ldb (elements)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vit_prisma_amd import _native as N
L = N.lib(); dev = torch.device("cuda:0"); st = torch.cuda.current_stream().cuda_stream
shapes = [("qkv", 25600, 2304, 768), ("oproj", 25600, 768, 768), ("mlp1", 25600, 3072, 768), ("mlp2", 25600, 768, 3072), ("sq4096", 4096, 4096, 4096),
          ("sq8192", 8192, 8192, 8192)]
for name, M, Nn, K in shapes:
    row = []
    for pa, pb in [(0, 0), (64, 0), (0, 64), (64, 64), (32, 32), (8, 8), (192, 192)]:
        A = torch.randn(M, K + pa, device=dev).bfloat16(); B = (torch.randn(Nn, K + pb, device=dev) * 0.05).bfloat16()
        bias = torch.randn(Nn, device=dev).bfloat16(); C = torch.empty(M, Nn, device=dev, dtype=torch.bfloat16)
        call = lambda: N.check(L.pv_gemm_bias(1, A.data_ptr(), K + pa, B.data_ptr(), K + pb, bias.data_ptr(), C.data_ptr(), Nn, M, Nn, K, st), "gemm")
        for _ in range(3): call()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): call()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        row.append(f"pad({pa},{pb}) {us:6.1f}us {2.0 * M * Nn * K / us / 1e6:5.0f}TF")
    print(f"{name:7s}: " + " | ".join(row), flush=True)
