# select kernel with two waves per token (16 tokens resident per CU) against four: targeted tests, kernel stats A/B, SAE bench legs
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4n; rm -rf $O; mkdir -p $O
cd $R
T0=$(date +%s)
timeout 500 python -m pytest tests/test_native_sae_gpu.py -m gpu -q -p no:cacheprovider --timeout=300 -k "native_step_vs_oracle or filtered_encoder or ties or reproducible or config3 or feature_parallel_world2" > $O/t_sae.log 2>&1; echo "sae tests rc=$? $(( $(date +%s) - T0 ))s"; tail -3 $O/t_sae.log
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
  PV_TUNE=sel_wide=$v timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p$v -o s -- python $R/tools/prof_sae.py > $O/prof$v.out 2> $O/prof$v.err
  cp $O/p$v/s_kernel_stats.csv $O/sel_wide${v}_kernel_stats.csv; rm -rf $O/p$v
  echo "sel_wide=$v"; grep -h 'sae_select_kernel' $O/sel_wide${v}_kernel_stats.csv | awk -F'",' '{print $2}' | cut -d, -f1-3; grep -o "'ms_per_step': [0-9.]*" $O/prof$v.out | head -1
done
echo "total $(( $(date +%s) - T0 ))s"
