R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3p; rm -rf $O; mkdir -p $O
cd $R
timeout 300 python tools/gemm_stride_probe.py > $O/stride.log 2>&1; echo "rc=$?" >> $O/stride.log
cat $O/stride.log | cut -c1-400
