#!/bin/bash
# round 5: ghost gradients with tokens sharded over ranks -- shard-sum tests, the trainer's path over RCCL at world 1, every ghost test
set -x
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_native_sae_gpu.py -q -m gpu -k "ghost or rccl_world1" 2>&1 | tail -30 > gpurun_out/r5j_pytest.txt
cat gpurun_out/r5j_pytest.txt
