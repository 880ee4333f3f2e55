python -m pytest tests/test_native_vit_gpu.py -m gpu -x -q -k "within_reference_bf16_budget or full_size or attention_core" 2>&1 | tail -3
for r in 1 2; do
for v in half wave; do
  e=A=1; [ $v = wave ] && e=PV_LN_WAVE=1
  env $e python bench.py --no-sae --no-l14 --no-cpu-baseline --steps 30 --warmup 8 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v bench', j['value'], j['ms_per_step'], j['kernels']['layernorm']['avg_launch_us'], j['kernels']['attention']['avg_launch_us'])"
done; done
