R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3v; rm -rf $O; mkdir -p $O
cd $R
timeout 600 python tools/attn_l14_time.py > $O/attn.json 2> $O/err.log; echo "rc=$?"
cat $O/attn.json; tail -3 $O/err.log
timeout 600 python -m pytest tests/test_native_vit_gpu.py -m gpu -q -k "long_sequence or l14 or L14" > $O/t.log 2>&1; tail -3 $O/t.log
