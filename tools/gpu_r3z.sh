R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3z; rm -rf $O; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_native_vit_gpu.py -m gpu -q -x -k "long_sequence" > $O/t.log 2>&1; echo "rc=$?" >> $O/t.log
tail -25 $O/t.log | cut -c1-300
timeout 600 python tools/attn_l14_time.py > $O/attn.json 2> $O/err.log; echo "rc=$?"
cat $O/attn.json; tail -3 $O/err.log
