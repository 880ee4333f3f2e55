R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4l; rm -rf $O; mkdir -p $O
cd $R
timeout 400 python -m pytest tests/test_native_vit_gpu.py -m gpu -q -p no:cacheprovider --timeout=200 -k "flag_gated or hooked_sae_vit or fallback_dispatch or edge_stage or mutating" > $O/t_vit.log 2>&1; echo "rc=$?"; tail -5 $O/t_vit.log
