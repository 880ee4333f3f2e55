#!/bin/bash
# round 5: the gated step's sparse form with ONE decode gather and the dual-term backward -- parity, time, kernel profile
set -x
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_native_sae_gpu.py -x -q -m gpu -k "gated or relu" 2>&1 | tail -15 > gpurun_out/r5f_pytest.txt
cat gpurun_out/r5f_pytest.txt
VARIANT=gated_relu_l0_64 python tools/sae_variant_time.py > gpurun_out/r5f_gated_l0_64.json 2> gpurun_out/r5f_err.txt
cat gpurun_out/r5f_gated_l0_64.json
( cd /tmp && VARIANT=gated_relu_l0_64 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_g -o gated -- python $R/tools/sae_variant_time.py > /tmp/prof_g.out 2>&1 )
f=$(find /tmp/prof_g -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && head -45 "$f" > gpurun_out/r5f_gated_l0_64_kernel_stats.csv
