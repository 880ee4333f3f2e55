# K-loop A/B incl. the full-line form + the tests that changed
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2d; rm -rf $O; mkdir -p $O
cd $R
timeout 300 python tools/gemm_ab.py 0,1,2 > $O/gemm_ab.log 2>&1; echo "rc=$?" >> $O/gemm_ab.log; cat $O/gemm_ab.log
for lp in -1 2 -1 2; do
  timeout 200 python bench.py --no-sae --no-l14 --no-cpu-baseline --allow-overrides --tune gemm_loop=$lp > $O/b32.json 2> $O/b32.err
  python -c "
import json; d=json.load(open('$O/b32.json')); print('loop$lp', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['avg_launch_us'])" 2>&1 | tee -a $O/summary.log
done
timeout 900 python -m pytest tests/test_native_vit_gpu.py tests/test_native_sae_gpu.py -m gpu -q -x -k "ragged or do_not_depend or sparse_gradient" > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log; tail -4 $O/tests.log
