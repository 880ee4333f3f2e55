"""A/B of the K-loop forms of the one-workgroup-per-CU bf16 GEMM (tuning key gemm_loop): bit-equality against
form 0 on the B/32 bs=512 shapes (+ a ragged M) and time per launch for every epilogue the forward uses.
    python tools/gemm_ab.py [loops, default 0,1,2]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vit_prisma_amd import _native as N

L = N.lib()
dev = torch.device("cuda:0")
loops = [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else "0,1,2".split(","))]
reps = int(os.environ.get("REPS", "30"))
shapes = [("qkv", 25600, 2304, 768), ("oproj", 25600, 768, 768), ("mlp1", 25600, 3072, 768), ("mlp2", 25600, 768, 3072),
          ("ragged", 25600 - 37, 1024, 1024), ("sq4096", 4096, 4096, 4096)]
st = torch.cuda.current_stream().cuda_stream
torch.manual_seed(0)
for name, M, Nn, K in shapes:
    A = torch.randn(M, K, device=dev).bfloat16(); B = (torch.randn(Nn, K, device=dev) * 0.05).bfloat16()
    bias = torch.randn(Nn, device=dev).bfloat16()
    ref = None
    row = []
    for lp in loops:
        N.set_tuning("gemm_loop", lp)
        C = torch.zeros(M, Nn, device=dev, dtype=torch.bfloat16)
        for _ in range(3):
            N.check(L.pv_gemm_bias(1, A.data_ptr(), K, B.data_ptr(), K, bias.data_ptr(), C.data_ptr(), Nn, M, Nn, K, st), "gemm")
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            L.pv_gemm_bias(1, A.data_ptr(), K, B.data_ptr(), K, bias.data_ptr(), C.data_ptr(), Nn, M, Nn, K, st)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        if ref is None:
            ref = C.clone(); same = "ref"
        else:
            same = "bit-identical" if torch.equal(ref.view(torch.int16), C.view(torch.int16)) else \
                f"DIFFERS max|d|={float((ref.float() - C.float()).abs().max()):.3e}"
        row.append(f"loop{lp}: {us:7.1f} us {2.0 * M * Nn * K / us / 1e6:6.0f} TF [{same}]")
    print(f"{name:7s} {M}x{Nn}x{K}: " + " | ".join(row), flush=True)
N.set_tuning("reset")
