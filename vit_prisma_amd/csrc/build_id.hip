// pv_build_id(): the hash of the sources this binary was built from (vit_prisma_amd/build.py writes _obj/build_id.inc before
// compiling: sha256 over csrc/*.hip, csrc/*.hpp and include/pv_native.h, names and contents, sorted).  libpvnative.so is
// git-ignored and travels to a GPU box as a prebuilt file next to the sources; vit_prisma_amd/_native.py recomputes the
// hash from the sources it finds there and tests/test_native_abi_cpu.py / the GPU suite assert that the two agree.
#include "../../include/pv_native.h"
#include "_obj/build_id.inc"

extern "C" const char* pv_build_id(void) { return PV_BUILD_ID; }
