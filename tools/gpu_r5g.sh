#!/bin/bash
# round 5: the top-k form of the gated step -- parity, time
set -x
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_native_sae_gpu.py -x -q -m gpu -k "gated" 2>&1 | tail -25 > gpurun_out/r5g_pytest.txt
cat gpurun_out/r5g_pytest.txt
