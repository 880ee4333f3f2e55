R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3j; rm -rf $O; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_native_sae_gpu.py -m gpu -q -k "checkpoint_mid_run" > $O/t.log 2>&1; echo "rc=$?" >> $O/t.log
tail -12 $O/t.log | cut -c1-300
