# K-loop ablations of the split-fp16 dense GEMM (tools/build_variant.sh builds with -DPV_DG_NOLOAD / NOMFMA / NOCVT; results are wrong by
# construction, only the step time is read): which part of the loop the 3.9 ms of the ReLU + L1 step at the published L0 is made of.
R=$GRAFT_REPO_ROOT; cd $R
for v in ${VARIANTS:-base dg_NOLOAD dg_NOMFMA dg_HALFCVT}; do
  if [ $v = base ]; then unset PV_NATIVE_LIB; unset PV_ALLOW_STALE_LIB; else export PV_NATIVE_LIB=tools/variants/libpvnative_$v.so PV_ALLOW_STALE_LIB=1; fi
  echo -n "$v: "; STEPS=6 timeout 120 python tools/prof_relu_dense.py 2>/dev/null | tail -1 | cut -c1-120
done
