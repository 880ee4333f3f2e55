"""The B/32 bs=512 GEMM shapes on this package's kernel beside the vendor library's (torch.matmul / F.linear = hipBLASLt / rocBLAS),
as a yardstick only: nothing under vit_prisma_amd/ calls the library.  Run under rocprofv3 --kernel-trace --stats to see which library
kernels (macro tile, MFMA shape in their names) the heuristics pick.  Inputs: N(0,1) bf16, weights [N][K] K-contiguous."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from vit_prisma_amd import _native as N
L = N.lib()
dev = torch.device("cuda:0")
shapes = [("qkv", 25600, 2304, 768), ("oproj", 25600, 768, 768), ("mlp1", 25600, 3072, 768), ("mlp2", 25600, 768, 3072), ("sq8192", 8192, 8192, 8192)]
reps = int(os.environ.get("REPS", "30"))


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


for name, M, Nn, K in shapes:
    A = torch.randn(M, K, device=dev).bfloat16(); B = torch.randn(Nn, K, device=dev).bfloat16()
    bias = torch.randn(Nn, device=dev).bfloat16(); C = torch.empty(M, Nn, device=dev, dtype=torch.bfloat16)
    Bt = B.t().contiguous()
    st = torch.cuda.current_stream().cuda_stream
    ours = timed(lambda: L.pv_gemm_bias(1, A.data_ptr(), K, B.data_ptr(), K, bias.data_ptr(), C.data_ptr(), Nn, M, Nn, K, st))
    lib_nt = timed(lambda: F.linear(A, B, bias))               # weight [N][K]: the layout this package's kernel reads
    lib_nn = timed(lambda: torch.addmm(bias, A, Bt))           # weight [K][N]: the reference's own parameter layout
    lib_nobias = timed(lambda: torch.matmul(A, B.t()))
    fl = 2.0 * M * Nn * K / 1e6
    print(f"{name:8s} {M}x{Nn}x{K}: ours {ours:7.1f} us {fl/ours:7.1f} TF | library NT+bias {lib_nt:7.1f} us {fl/lib_nt:7.1f} TF | "
          f"NN+bias {lib_nn:7.1f} us {fl/lib_nn:7.1f} TF | NT no bias {lib_nobias:7.1f} us {fl/lib_nobias:7.1f} TF", flush=True)
