#!/bin/bash
# round 5: time + kernel profile of the top-k gated step
set -x
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
VARIANT=gated_topk python tools/sae_variant_time.py > gpurun_out/r5h_gated_topk.json 2> gpurun_out/r5h_err.txt
cat gpurun_out/r5h_gated_topk.json; tail -3 gpurun_out/r5h_err.txt
( cd /tmp && VARIANT=gated_topk rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_g -o gated -- python $R/tools/sae_variant_time.py > /tmp/prof_g.out 2>&1 )
f=$(find /tmp/prof_g -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && head -45 "$f" > gpurun_out/r5h_gated_topk_kernel_stats.csv
