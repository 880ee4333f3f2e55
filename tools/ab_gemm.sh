for v in base nob; do
  lib=$PWD/tools/variants/libpvnative_$v.so; [ $v = base ] && lib=$PWD/vit_prisma_amd/libpvnative.so
  for t in 5 4; do for d in 2 3; do
    echo "== $v tile=$t dbg=$d"; PV_NATIVE_LIB=$lib PV_GEMM_DBG=$d PV_GEMM_TILE=$t REPS=20 python tools/gemm_bench.py 2>&1 | grep -v amdgpu.ids | awk '{printf "%s %s us %s TF | ", $1, $3, $5} END {print ""}'
  done; done
done
