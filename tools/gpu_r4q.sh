R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4q; rm -rf $O; mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_native_vit_gpu.py -m gpu -q -p no:cacheprovider --timeout=200 -k "hooked_sae_vit" > $O/t.log 2>&1; echo "rc=$?"; tail -6 $O/t.log
