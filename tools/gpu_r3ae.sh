R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3ae; rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_native_sae_gpu.py -m gpu -q -x -k "gated or transcoder or token_shards or relu" > $O/t.log 2>&1; echo "rc=$?" >> $O/t.log
tail -15 $O/t.log | cut -c1-300
