R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3y; rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_native_vit_gpu.py -m gpu -q -x -k "hooks or substitution or mutating" > $O/t.log 2>&1; echo "rc=$?" >> $O/t.log
tail -40 $O/t.log | cut -c1-300
