# ViT GPU tests + the L/14 leg with kernel stats
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/vit; rm -rf $O; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_native_vit_gpu.py -m gpu -q > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
cd /tmp && export TMPDIR=/tmp
STEPS=6 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_l14 -o l14 -- python $R/tools/l14_run.py > $O/l14.json 2> $O/l14.err
cp $O/prof_l14/l14_kernel_stats.csv $O/l14_kernel_stats.csv; rm -rf $O/prof_l14
cd $R; tail -15 $O/tests.log; cat $O/l14.json; head -12 $O/l14_kernel_stats.csv | cut -c1-160
