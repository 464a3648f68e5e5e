# usage (GPU box): bash tools/r6/distill_round.sh <tag>  -- static distillation: tests + the distill bench leg (+ its kernel timeline)
TAG=${1:-r6f}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/$TAG
mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_matcher.py tests/test_gpu_distill.py tests/test_gpu_distill_fullsize.py -m gpu -q -p no:cacheprovider ) > $O/pytest.log 2>&1
grep -v amdgpu.ids $O/pytest.log | tail -30 | cut -c1-300
( timeout 600 python bench.py --distill --batch 4 --no-cpu-baseline --no-roofline ) > $O/bench_distill.log 2>&1
grep "metric\|\[bench\]" $O/bench_distill.log | cut -c1-1100
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o d -- python bench.py --distill --batch 4 --no-cpu-baseline --no-roofline > $O/bench_rocprof.log 2>&1
python tools/timeline.py $O/prof/d_kernel_trace.csv $O/timeline.txt $O/sequence.txt > /dev/null 2>&1
rm -rf $O/prof
head -14 $O/timeline.txt | cut -c1-160
