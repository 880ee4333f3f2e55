# rocprofv3 kernel stats of tools/prof_sae.py (PV_TUNE passes through) -> gpurun_out/prof_sae_once_kernel_stats.csv
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 250 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_once -o p -- python $R/tools/prof_sae.py > $O/prof_sae_once.log 2>&1
cp $(find $O/prof_once -name '*kernel_stats.csv' | head -1) $O/prof_sae_once_kernel_stats.csv
rm -rf $O/prof_once
