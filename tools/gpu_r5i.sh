#!/bin/bash
# round 5: SAE steps at d_in up to 1280 (ViT-H/14's width) -- parity on every step form
set -x
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_native_sae_gpu.py -q -m gpu -k "1280 or 1156" 2>&1 | tail -30 > gpurun_out/r5i_pytest.txt
cat gpurun_out/r5i_pytest.txt
