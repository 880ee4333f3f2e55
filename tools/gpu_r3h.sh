# round 3, eighth GPU pass: hooks inside the block on the split plan + the whole ViT suite
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3h; rm -rf $O; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_native_vit_gpu.py -m gpu -q -x -k "hooks" > $O/hook_tests.log 2>&1; echo "rc=$?" >> $O/hook_tests.log
tail -30 $O/hook_tests.log
