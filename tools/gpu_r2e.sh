# default = full-line loop: full GPU tests, SAE numbers, LP2 vs LP3 (all pieces behind the last k-step)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2e; rm -rf $O; mkdir -p $O
cd $R
timeout 300 python tools/gemm_ab.py 1,2,3 > $O/gemm_ab.log 2>&1; echo "rc=$?" >> $O/gemm_ab.log; cat $O/gemm_ab.log
for lp in 2 3 2 3; do
  timeout 200 python bench.py --no-sae --no-l14 --no-cpu-baseline --allow-overrides --tune gemm_loop=$lp > $O/b32.json 2> $O/b32.err
  python -c "
import json; d=json.load(open('$O/b32.json')); print('loop$lp', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['avg_launch_us'])" 2>&1 | tee -a $O/summary.log
done
timeout 400 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; tail -c 200 $O/bench.err
python - <<PY
import json
d=json.load(open('$O/bench.json'))
print('b32', d['value'], d['ms_per_step'], 'gemm', d['roofline']['achieved'])
print('sae', d['sae']['value'], d['sae']['ms_per_step'], d['sae']['kernels'], 'e2e', d['sae']['end_to_end']['value'])
print('l14', d['l14_336_pattern']['value'], d['l14_336_pattern']['ms_per_step'], d['l14_336_pattern']['roofline'])
PY
timeout 1200 python -m pytest tests/ -m gpu -q > $O/tests_gpu.log 2>&1; echo "tests rc=$?" >> $O/tests_gpu.log; tail -4 $O/tests_gpu.log
