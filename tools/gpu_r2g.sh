# attention (T <= 64): swapped-operand form (8-byte LDS row writes, exact reciprocal scale, per-row NaN test) vs the committed kernel
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2g; rm -rf $O; mkdir -p $O
cd $R
for v in new old new old; do
  if [ $v = old ]; then export PV_NATIVE_LIB=$R/tools/variants/libpvnative_attnhead.so; else unset PV_NATIVE_LIB; fi
  timeout 200 python bench.py --no-sae --no-l14 --no-cpu-baseline --allow-overrides > $O/b32.json 2> $O/b32.err
  python -c "
import json; d=json.load(open('$O/b32.json')); print('$v', d['value'], d['ms_per_step'], d['kernels']['attention']['avg_launch_us'])" 2>&1 | tee -a $O/summary.log
done
unset PV_NATIVE_LIB
timeout 900 python -m pytest tests/test_native_vit_gpu.py -m gpu -q -x > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log; tail -4 $O/tests.log
