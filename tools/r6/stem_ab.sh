TAG=${1:-r6j}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/$TAG
mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_model.py -m gpu -q -p no:cacheprovider -k "stem" ) 2>&1 | grep -v amdgpu.ids | tail -3
( timeout 300 python tools/r3/stem_bench.py ) 2>&1 | grep -v amdgpu.ids | tail -6
if [ -f build/libtoist_hip_prev.so ]; then ( TOIST_HIP_LIB=build/libtoist_hip_prev.so timeout 300 python tools/r3/stem_bench.py ) 2>&1 | grep -v amdgpu.ids | tail -6; fi
