R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3x; rm -rf $O; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_native_sae_gpu.py -m gpu -q -x > $O/t.log 2>&1; echo "rc=$?" >> $O/t.log
tail -5 $O/t.log | cut -c1-300
