export PV_GEMM_V7=5 WHICH=2
echo "== bs128"; BS=128 python tools/gemm_trace.py 2>&1 | grep -v amdgpu.ids
echo "== bs256"; BS=256 python tools/gemm_trace.py 2>&1 | grep -v amdgpu.ids
echo "== bs512 nt pre"; PV_GEMM_DBG=32 python tools/gemm_trace.py 2>&1 | grep -v amdgpu.ids
echo "== bs512 nt pre+post"; PV_GEMM_DBG=96 python tools/gemm_trace.py 2>&1 | grep -v amdgpu.ids
echo "== bs512 base"; python tools/gemm_trace.py 2>&1 | grep -v amdgpu.ids
