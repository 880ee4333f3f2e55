# attention (T <= 64) with whole-row Q / K staging vs direct fragment loads
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2f; rm -rf $O; mkdir -p $O
cd $R
for ad in 0 1 0 1; do
  timeout 200 python bench.py --no-sae --no-l14 --no-cpu-baseline --allow-overrides --tune attn_direct=$ad > $O/b32.json 2> $O/b32.err
  python -c "
import json; d=json.load(open('$O/b32.json')); print('attn_direct$ad', d['value'], d['ms_per_step'], d['kernels']['attention'])" 2>&1 | tee -a $O/summary.log
done
timeout 900 python -m pytest tests/test_native_vit_gpu.py -m gpu -q -x > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log; tail -4 $O/tests.log
