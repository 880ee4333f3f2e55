"""What the HIP runtime says about the residency of the filtered encoder's small kernels (workgroups per CU of sae_select_kernel<3>,
sae_thr_kernel<32>, relu_select_kernel<3>; VGPRs of the first) -- beside tools/probes/select_timeline.py, which counts them alive."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vit_prisma_amd import _native as N
torch.zeros(1, device="cuda")
out = (C.c_int32 * 4)()
print("rc", N.lib().pv_debug_select_occupancy(out), "workgroups per CU: select", out[0], "thr", out[1], "relu_select", out[2], "| select VGPRs", out[3])
