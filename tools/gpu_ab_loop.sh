# K-loop A/B (tuning key gemm_loop) + the GPU test suite
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ab; rm -rf $O; mkdir -p $O
cd $R
timeout 300 python tools/gemm_ab.py ${LOOPS:-0,1,2,3} > $O/gemm_ab.log 2>&1; echo "rc=$?" >> $O/gemm_ab.log
for lp in ${BENCH_LOOPS:-0 1 3}; do
  timeout 200 python bench.py --no-sae --no-l14 --no-cpu-baseline --allow-overrides --tune gemm_loop=$lp > $O/b32_loop$lp.json 2> $O/b32_loop$lp.err
  python -c "
import json; d=json.load(open('$O/b32_loop$lp.json')); print('loop$lp', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['avg_launch_us'])" >> $O/summary.log 2>&1
done
if [ -z "$SKIP_TESTS" ]; then
timeout 1200 python -m pytest tests/ -m gpu -q -x > $O/tests_gpu.log 2>&1; echo "tests rc=$?" >> $O/tests_gpu.log
fi
cat $O/gemm_ab.log; cat $O/summary.log; tail -5 $O/tests_gpu.log
