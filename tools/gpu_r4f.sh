# library yardstick for the four GEMM shapes (+ kernel names through rocprofv3), and the re-run of the one stale test
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4f; rm -rf $O; mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_native_vit_gpu.py -m gpu -q -p no:cacheprovider --timeout=200 -k "fallback_dispatch or edge_stage" > $O/t_vit.log 2>&1; echo "rc=$?"; tail -3 $O/t_vit.log
timeout 200 python tools/gemm_vs_library.py > $O/gemm_vs_library.txt 2>&1; cat $O/gemm_vs_library.txt
cd /tmp && export TMPDIR=/tmp
REPS=5 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o lib -- python $R/tools/gemm_vs_library.py > $O/prof.out 2>&1
cp $O/prof/lib_kernel_stats.csv $O/lib_kernel_stats.csv; rm -rf $O/prof
cut -c1-260 $O/lib_kernel_stats.csv | head -20
