# ViT GPU tests + B/32 bench line (no SAE / L14 / CPU legs) + the L/14 leg with kernel stats
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/vit; rm -rf $O; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_native_vit_gpu.py -m gpu -q > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
timeout 300 python bench.py --no-sae --no-l14 --no-cpu-baseline > $O/b32.json 2> $O/b32.err
cd /tmp && export TMPDIR=/tmp
STEPS=6 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_l14 -o l14 -- python $R/tools/l14_run.py > $O/l14.json 2> $O/l14.err
cp $O/prof_l14/l14_kernel_stats.csv $O/l14_kernel_stats.csv; rm -rf $O/prof_l14
cd $R; tail -5 $O/tests.log; python -c "
import json; d=json.load(open('$O/b32.json')); print('b32', d['value'], d['ms_per_step'], d['kernels']['attention'])"; cat $O/l14.json | cut -c1-200; head -4 $O/l14_kernel_stats.csv | cut -c1-160
