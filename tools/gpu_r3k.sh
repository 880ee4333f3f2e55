R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3k; rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_native_sae_gpu.py tests/test_native_vit_gpu.py -m gpu -q -k "relu or fallback_dispatch or library_is_loaded" > $O/t.log 2>&1; echo "rc=$?" >> $O/t.log
tail -25 $O/t.log | cut -c1-500
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
