"""HBM write / copy bandwidth reference points (torch fill / copy kernels) for the epilogue analysis."""
import torch
dev = torch.device("cuda:0")
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for mb in (39, 118, 314, 1024):
    n = mb * 1024 * 1024 // 2
    a = torch.empty(n, dtype=torch.bfloat16, device=dev); b = torch.randn(n, device=dev).bfloat16()
    s = t(lambda: a.zero_()); print(f"fill  {mb:5d} MB: {s*1e6:8.1f} us  {mb*1.048576e-3/s/1e3:6.2f} TB/s write")
    s = t(lambda: a.copy_(b)); print(f"copy  {mb:5d} MB: {s*1e6:8.1f} us  {2*mb*1.048576e-3/s/1e3:6.2f} TB/s r+w")
    s = t(lambda: b.sum()); print(f"read  {mb:5d} MB: {s*1e6:8.1f} us  {mb*1.048576e-3/s/1e3:6.2f} TB/s read")
