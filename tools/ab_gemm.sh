python -m pytest tests/test_native_vit_gpu.py -m gpu -x -q 2>&1 | tail -3
for r in 1 2; do
for v in v8 v7; do
  e=A=1; [ $v = v7 ] && e=PV_GEMM_NO_V8=1
  env $e python bench.py --no-sae --no-l14 --no-cpu-baseline --steps 30 --warmup 8 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v bench', j['value'], j['ms_per_step'], j['roofline']['achieved'], j['roofline']['avg_launch_us'])"
done; done
python tools/gemm_trace.py 2>&1 | grep -v "amdgpu.ids\|timeline\|avg resident"
