# nontemporal stores for the tap-only streams (GEMM pre / attn_out / mlp_out taps, LayerNorm's fp32 hook_normalized, attention scores / pattern):
# A/B against a -DPV_NO_NT build of the same sources on one box: B/32 main line, L/14 leg; then the ViT GPU tests on the product library
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4o; rm -rf $O; mkdir -p $O
cd $R
T0=$(date +%s)
for v in plain stream plain stream; do
  L=$R/vit_prisma_amd/libpvnative.so; [ $v = plain ] && L=$R/tools/variants/libpvnative_nont.so
  PV_NATIVE_LIB=$L timeout 200 python bench.py --no-sae --no-l14 --no-cpu-baseline --allow-overrides --steps 30 --warmup 5 > $O/$v.json 2> $O/$v.err
  python - <<PY
import json
d=json.loads([l for l in open('$O/$v.json') if l.startswith('{"metric"')][0])
print('$v', d['value'], d['ms_per_step'], d['roofline']['frac'], {k:v['avg_launch_us'] for k,v in d['roofline']['instances'].items()}, flush=True)
PY
done
for v in plain stream; do
  L=$R/vit_prisma_amd/libpvnative.so; [ $v = plain ] && L=$R/tools/variants/libpvnative_nont.so
  PV_NATIVE_LIB=$L STEPS=6 timeout 200 python tools/l14_run.py > $O/l14_$v.json 2> $O/l14_$v.err; echo "l14 $v $(grep -o '"value": [0-9.]*' $O/l14_$v.json | head -1) $(grep -o '"ms_per_step": [0-9.]*' $O/l14_$v.json | head -1)"
done
echo "ab done $(( $(date +%s) - T0 ))s"
timeout 500 python -m pytest tests/test_native_vit_gpu.py -m gpu -q -p no:cacheprovider --timeout=300 > $O/t_vit.log 2>&1; echo "vit tests rc=$? $(( $(date +%s) - T0 ))s"; tail -3 $O/t_vit.log
