R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3q; rm -rf $O; mkdir -p $O
cd $R
timeout 120 ./tools/probes/l2_to_cu_probe.bin > $O/probe.log 2>&1; echo "rc=$?" >> $O/probe.log
cat $O/probe.log
