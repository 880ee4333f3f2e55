# full-line loop: slab 1's A pieces issued before the wait for slab 0 (new) vs after the first barrier (old = HEAD build)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2h; rm -rf $O; mkdir -p $O
cd $R
REPS=20 timeout 200 python tools/gemm_ab.py 0,2 > $O/gemm_ab.log 2>&1; cat $O/gemm_ab.log | grep -v amdgpu
for v in new old new old; do
  if [ $v = old ]; then export PV_NATIVE_LIB=$R/tools/variants/libpvnative_head.so; else unset PV_NATIVE_LIB; fi
  timeout 200 python bench.py --no-l14 --no-cpu-baseline --allow-overrides --steps 30 > $O/b32.json 2> $O/b32.err
  python -c "
import json; d=json.load(open('$O/b32.json')); print('$v', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], 'sae', d['sae']['ms_per_step'], d['sae']['kernels']['encode_topk']['avg_us'])" 2>&1 | tee -a $O/summary.log
done
