# usage (GPU box): bash tools/r6/lsap_round.sh <tag>  -- LSAP: tests, timing (new / previous build), the distillation leg
TAG=${1:-r6h}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/$TAG
mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_matcher.py tests/test_gpu_distill.py tests/test_gpu_distill_fullsize.py -m gpu -q -p no:cacheprovider ) > $O/pytest.log 2>&1
grep -v amdgpu.ids $O/pytest.log | tail -6 | cut -c1-300
( timeout 300 python tools/r6/lsap_bench.py ) 2>&1 | grep -v amdgpu.ids | tee $O/lsap_new.txt
if [ -f build/libtoist_hip_prev.so ]; then ( TOIST_HIP_LIB=build/libtoist_hip_prev.so timeout 300 python tools/r6/lsap_bench.py ) 2>&1 | grep -v amdgpu.ids | tee $O/lsap_old.txt; fi
( timeout 600 python bench.py --distill --batch 4 --no-cpu-baseline --no-roofline ) > $O/bench_distill.log 2>&1
grep "metric\|\[bench\]" $O/bench_distill.log | cut -c1-200
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o d -- python bench.py --distill --batch 4 --no-cpu-baseline --no-roofline > $O/bench_rocprof.log 2>&1
python tools/timeline.py $O/prof/d_kernel_trace.csv $O/timeline.txt $O/sequence.txt > /dev/null 2>&1
rm -rf $O/prof
head -8 $O/timeline.txt | cut -c1-160
