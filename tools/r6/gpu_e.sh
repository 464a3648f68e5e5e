# usage (GPU box): bash tools/r6/gpu_e.sh <tag>  -- LSAP: tests + timing, old library beside the new one when build/libtoist_hip_prev.so exists
TAG=${1:-r6e}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/$TAG
mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_matcher.py tests/test_gpu_distill.py tests/test_gpu_distill_fullsize.py -m gpu -q -p no:cacheprovider ) > $O/pytest.log 2>&1
grep -v amdgpu.ids $O/pytest.log | tail -12 | cut -c1-300
( timeout 300 python tools/r6/lsap_bench.py ) 2>&1 | grep -v amdgpu.ids | tee $O/lsap_new.txt
if [ -f build/libtoist_hip_prev.so ]; then ( TOIST_HIP_LIB=build/libtoist_hip_prev.so timeout 300 python tools/r6/lsap_bench.py ) 2>&1 | grep -v amdgpu.ids | tee $O/lsap_old.txt; fi
( timeout 600 python bench.py --distill --batch 4 --no-cpu-baseline --no-roofline ) > $O/bench_distill.log 2>&1
grep metric $O/bench_distill.log | cut -c1-900
