R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3o; rm -rf $O; mkdir -p $O
cd $R
timeout 600 python tools/gemm_v8_ab.py > $O/ab.log 2>&1; echo "rc=$?" >> $O/ab.log
cat $O/ab.log | cut -c1-420 | tail -60
