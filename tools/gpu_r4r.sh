# streamed Adam state in the tile-walk W_enc kernel too (engines that keep the parameter's own layout: transcoder, gated, multi-rank):
# parity of those steps, then step times against a -DPV_NO_NT build
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4r; rm -rf $O; mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_native_sae_gpu.py -m gpu -q -p no:cacheprovider --timeout=200 -k "transcoder_steps_vs_oracle or gated_step_vs_oracle or rccl or dp_world2" > $O/t.log 2>&1; echo "rc=$?"; tail -3 $O/t.log
for v in plain stream; do
  L=$R/vit_prisma_amd/libpvnative.so; [ $v = plain ] && L=$R/tools/variants/libpvnative_saeplain.so
  PV_NATIVE_LIB=$L timeout 200 python tools/variants_time.py > $O/var_$v.txt 2>&1; echo $v; tail -8 $O/var_$v.txt | cut -c1-200
done
