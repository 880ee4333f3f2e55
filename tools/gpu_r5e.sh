#!/bin/bash
# round 5: kernel profile of the gated step in the sparse regime (L0 ~ 64)
set -x
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( cd /tmp && VARIANT=gated_relu_l0_64 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_g -o gated -- python $R/tools/sae_variant_time.py > /tmp/prof_g.out 2>&1 )
f=$(find /tmp/prof_g -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && head -45 "$f" > gpurun_out/r5e_gated_l0_64_kernel_stats.csv
tail -3 /tmp/prof_g.out
