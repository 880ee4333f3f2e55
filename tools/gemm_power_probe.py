"""Is the full-chip K loop slower than a few-CU K loop because of the clock (power), or because of shared bandwidth?  The QKV-shaped
GEMM (25600 x 2304 x 768, bias epilogue) on random / small-integer / zero operands, with the shader clock and socket power
sampled from rocm-smi while it loops."""
import os, subprocess, sys, threading, time, re
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vit_prisma_amd import _native as N
L = N.lib(); dev = torch.device("cuda:0"); st = torch.cuda.current_stream().cuda_stream
M, Nn, K = 25600, 2304, 768
samples = []
stop = False
def sampler():
    while not stop:
        try:
            o = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=5).stdout
            sclk = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", o)
            pw = re.search(r"Power \(W\): ([\d.]+)", o)
            samples.append((time.time(), int(sclk.group(1)) if sclk else -1, float(pw.group(1)) if pw else -1.0))
        except Exception as e:
            samples.append((time.time(), -2, -2.0))
th = threading.Thread(target=sampler); th.start()
for fill in ("random", "zeros", "smallint", "random"):
    if fill == "random":
        A = torch.randn(M, K, device=dev).bfloat16(); B = (torch.randn(Nn, K, device=dev) * 0.05).bfloat16()
    elif fill == "zeros":
        A = torch.zeros(M, K, device=dev).bfloat16(); B = torch.zeros(Nn, K, device=dev).bfloat16()
    else:
        A = torch.randint(0, 3, (M, K), device=dev).bfloat16(); B = torch.randint(0, 3, (Nn, K), device=dev).bfloat16()
    bias = torch.zeros(Nn, device=dev).bfloat16(); C = torch.empty(M, Nn, device=dev, dtype=torch.bfloat16)
    def go(n):
        for _ in range(n):
            L.pv_gemm_bias(1, A.data_ptr(), K, B.data_ptr(), K, bias.data_ptr(), C.data_ptr(), Nn, M, Nn, K, st)
    go(20); torch.cuda.synchronize()
    n0 = len(samples)
    t0 = time.time()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); go(30000); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 30000
    ss = [s for s in samples[n0:] if s[1] > 0]
    clk = sorted(s[1] for s in ss); pw = sorted(s[2] for s in ss)
    print(f"{fill:9s}: {us:7.1f} us/launch {2.0*M*Nn*K/us/1e6:7.0f} TFLOP/s over {time.time()-t0:.1f} s | sclk samples {len(clk)} median {clk[len(clk)//2] if clk else -1} MHz min {clk[0] if clk else -1} | power median {pw[len(pw)//2] if pw else -1} W", flush=True)
    time.sleep(1.0)
stop = True; th.join()
print("idle-ish samples:", samples[:2], samples[-2:])
