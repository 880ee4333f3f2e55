#!/bin/bash
# round 5: the tightened parity edges (bf16 flagged budget at B/32, HookedSAEViT at B/32, derived ghost tolerances)
mkdir -p gpurun_out; cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_native_vit_gpu.py -x -q -k "flag_gated_hook_points_at_b32 or hooked_sae_vit" > gpurun_out/r5l_parity_vit.txt 2>&1
tail -5 gpurun_out/r5l_parity_vit.txt
timeout 2400 python -m pytest tests/test_native_sae_gpu.py -x -q -k "relu_l1_dense_step_vs_oracle or topk_ghost_step_vs_oracle or store_taps" > gpurun_out/r5l_parity_sae.txt 2>&1
tail -8 gpurun_out/r5l_parity_sae.txt
