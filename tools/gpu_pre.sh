# input-pipeline kernel: tests + throughput of the transform at ImageNet-like sizes
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pre; rm -rf $O; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_native_transform_gpu.py -m gpu -q > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
timeout 300 python - > $O/rate.txt 2>&1 <<'P'
import torch, time
from vit_prisma_amd.transforms import GpuClipTransform
for (h, w) in ((375, 500), (1080, 1920)):
    x = torch.randint(0, 256, (256, h, w, 3), dtype=torch.uint8, device="cuda")
    for dt in (torch.bfloat16,):
        t = GpuClipTransform(224, device="cuda", dtype=dt)
        for _ in range(3): t(x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): t(x)
        torch.cuda.synchronize(); dtm = (time.perf_counter() - t0) / 10
        print(f"{h}x{w} -> 224 bf16: {256 / dtm:,.0f} images/s ({dtm * 1e3:.3f} ms per 256), native={t.last_native}")
        t.native = False
        for _ in range(3): t(x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): t(x)
        torch.cuda.synchronize(); dtm = (time.perf_counter() - t0) / 10
        print(f"   torch F.interpolate path: {256 / dtm:,.0f} images/s")
P
tail -8 $O/tests.log; cat $O/rate.txt
