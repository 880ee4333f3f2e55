R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3t; rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_native_sae_gpu.py -m gpu -q -k "gated" > $O/t.log 2>&1; echo "rc=$?" >> $O/t.log
tail -60 $O/t.log | cut -c1-300
