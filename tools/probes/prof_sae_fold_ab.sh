# per-kernel A/B of the top-k SAE step: folded launches (default) against single launches (PV_TUNE=sae_fold=0)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for v in 1 0; do
  PV_TUNE=sae_fold=$v timeout 250 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_fold_$v -o p -- python $R/tools/prof_sae.py > $O/prof_fold_$v.log 2>&1
  f=$(find $O/prof_fold_$v -name '*kernel_stats.csv' | head -1)
  cp $f $O/prof_fold_${v}_kernel_stats.csv
  rm -rf $O/prof_fold_$v
done
