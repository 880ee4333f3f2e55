# rocprofv3 kernel stats of tools/prof_relu_dense.py (PV_TUNE passes through) -> gpurun_out/prof_relu_dense_once_kernel_stats.csv
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
STEPS=16 timeout 250 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_rd_once -o p -- python $R/tools/prof_relu_dense.py > $O/prof_relu_dense_once.log 2>&1
cp $(find $O/prof_rd_once -name '*kernel_stats.csv' | head -1) $O/prof_relu_dense_once_kernel_stats.csv
rm -rf $O/prof_rd_once
