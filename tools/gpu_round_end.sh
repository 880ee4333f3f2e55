# round-end GPU pass: every -m gpu test, smoke(), then the evidence run
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/end; rm -rf $O; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/ -m gpu -q > $O/tests_gpu.log 2>&1; echo "tests rc=$?" >> $O/tests_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
tail -5 $O/tests_gpu.log; tail -2 $O/smoke.log
bash tools/final_profiles.sh
