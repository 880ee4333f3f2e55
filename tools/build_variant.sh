#!/bin/bash
# tools/build_variant.sh NAME [-DFLAG ...]: one source of csrc/ (SRC=gemm by default) rebuilt with extra defines into
# tools/variants/libpvnative_NAME.so (the other objects come from the regular in-tree build); select it with
# PV_NATIVE_LIB=tools/variants/libpvnative_NAME.so
set -e
cd "$(dirname "$0")/.."
name=$1; shift
src=${SRC:-gemm}
mkdir -p tools/variants
python -m vit_prisma_amd.build >/dev/null
# (SRC may list several sources: SRC="gemm rowops attention")
objs=$(ls vit_prisma_amd/csrc/_obj/*.o)
vobjs=""
for one in $src; do
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -ffp-contract=off -Iinclude "$@" -c vit_prisma_amd/csrc/$one.hip -o tools/variants/${one}_$name.o
  objs=$(echo "$objs" | grep -v "/$one.o")
  vobjs="$vobjs tools/variants/${one}_$name.o"
done
hipcc -shared -fPIC --offload-arch=gfx950 -o tools/variants/libpvnative_$name.so $objs $vobjs
rm $vobjs
echo built tools/variants/libpvnative_$name.so
