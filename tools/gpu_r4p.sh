# Adam moments / gradients streamed past the caches (nontemporal): A/B of the SAE legs against a -DPV_NO_NT build of sae.hip, kernel stats
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4p; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
T0=$(date +%s)
for v in plain stream plain stream; do
  L=$R/vit_prisma_amd/libpvnative.so; [ $v = plain ] && L=$R/tools/variants/libpvnative_saeplain.so
  PV_NATIVE_LIB=$L timeout 200 python $R/tools/prof_sae.py > $O/sae_$v.out 2> $O/sae_$v.err; echo "topk $v $(grep -o "'ms_per_step': [0-9.]*" $O/sae_$v.out | head -1)"
  PV_NATIVE_LIB=$L timeout 200 python $R/tools/prof_relu.py > $O/relu_$v.out 2> $O/relu_$v.err; echo "relu $v $(grep -o "'ms_per_step': [0-9.]*" $O/relu_$v.out | head -1)"
done
for v in plain stream; do
  L=$R/vit_prisma_amd/libpvnative.so; [ $v = plain ] && L=$R/tools/variants/libpvnative_saeplain.so
  PV_NATIVE_LIB=$L timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_$v -o s -- python $R/tools/prof_sae.py > /dev/null 2> $O/prof_$v.err
  cp $O/p_$v/s_kernel_stats.csv $O/${v}_kernel_stats.csv; rm -rf $O/p_$v
  echo $v; head -9 $O/${v}_kernel_stats.csv | awk -F'",' '{print substr($1,1,60), $2}' | cut -d, -f1-3
done
echo "total $(( $(date +%s) - T0 ))s"
