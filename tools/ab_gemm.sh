#!/bin/bash
# A/B of two environments on the B/32 bs=512 ViT leg, alternating, two repetitions each (run through gpurun: the
# numbers of ONE call on ONE box are comparable, boxes differ by several percent).
#   bash tools/ab_gemm.sh "PV_GEMM_TILE=0" "PV_GEMM_TILE=5"
A=${1:-A=1}; B=${2:-PV_GEMM_TILE=0}
for r in 1 2; do
  for e in "$A" "$B"; do
    env $e python bench.py --no-sae --no-l14 --no-cpu-baseline --steps 30 --warmup 8 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$e', j['value'], 'img/s', j['ms_per_step'], 'ms  gemm', j['roofline']['achieved'], 'TF', j['roofline']['avg_launch_us'], 'us  attn', j['kernels']['attention']['avg_launch_us'], 'ln', j['kernels']['layernorm']['avg_launch_us'])"
  done
done
