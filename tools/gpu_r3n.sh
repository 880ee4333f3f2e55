R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3n; rm -rf $O; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_native_sae_gpu.py -m gpu -q -x > $O/t.log 2>&1; echo "rc=$?" >> $O/t.log
tail -4 $O/t.log | cut -c1-300
timeout 600 python tools/tp_shard_times.py > $O/tp_shard_times.json 2> $O/tp.err
cat $O/tp_shard_times.json | tr -d '\n' | cut -c1-1500; echo; tail -3 $O/tp.err
