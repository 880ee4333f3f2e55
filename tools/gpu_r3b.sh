# round 3, second GPU pass: x64 SAEs (d_sae > 32768), the feature-parallel step with the merge kernel / bucket, its per-rank
# phase times at world 1/2/4/8, the diagnostic of the 768 -> 8192 failure of the first pass
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3b; rm -rf $O; mkdir -p $O
cd $R
timeout 300 python tools/tp_diag.py 768 8192 32 512 0 > $O/tp_diag.log 2>&1; tail -30 $O/tp_diag.log
PV_EXPERIMENTAL=1 timeout 1500 python -m pytest tests/test_native_sae_gpu.py -m gpu -q -x > $O/sae_tests.log 2>&1; echo "rc=$?" >> $O/sae_tests.log
tail -15 $O/sae_tests.log
timeout 600 python tools/tp_shard_times.py > $O/tp_shard_times.json 2> $O/tp_shard_times.err; cat $O/tp_shard_times.json; tail -3 $O/tp_shard_times.err
