#!/bin/bash
# round 5, first GPU pass of the persistent GEMM: A/B + traces
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 900 python tools/gemm_persist_ab.py > gpurun_out/r5a_persist_ab.txt 2>&1
echo "rc=$?" >> gpurun_out/r5a_persist_ab.txt
for p in 0 1; do
  PV_PERSIST=$p timeout 300 python tools/gemm_trace.py >> gpurun_out/r5a_trace.txt 2>&1
done
tail -40 gpurun_out/r5a_persist_ab.txt
