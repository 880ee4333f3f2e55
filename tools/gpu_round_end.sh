# what the driver runs at round end: every -m gpu test (stop at the first failure) and smoke()
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/round_end; rm -rf $O; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests/ -x -q -m gpu > $O/tests_gpu.log 2>&1; echo "tests rc=$?" >> $O/tests_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
tail -3 $O/tests_gpu.log; tail -2 $O/smoke.log
