# usage (GPU box): bash tools/r6/gpu_c.sh <tag>  -- re-run of fixed tests, default bench (kernel-trace roofline), attention blocks, xdec phases, replayed-step timeline + sequence
TAG=${1:-r6c}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/$TAG
mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_xdec.py tests/test_gpu_captured_step.py tests/test_gpu_segm.py -m gpu -q -p no:cacheprovider -s ) > $O/pytest.log 2>&1
grep -v amdgpu.ids $O/pytest.log | tail -15 | cut -c1-300
( time timeout 1200 python bench.py ) > $O/bench_default.log 2>&1
grep metric $O/bench_default.log | cut -c1-200
cp gpurun_out/bench_kernel_stats.csv $O/bench_kernel_stats.csv 2>/dev/null
( timeout 600 python tools/bench_attention.py ) 2>&1 | grep -v amdgpu.ids > $O/attention.txt
tail -5 $O/attention.txt
( timeout 300 python tools/r5/xdec_bench.py --train --bwd ) 2>&1 | grep -v amdgpu.ids > $O/xdec_bench.txt
grep -E "decoder forward|backward launch|one-launch|phase A" $O/xdec_bench.txt | cut -c1-300
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o $TAG -- python bench.py --no-cpu-baseline --no-roofline --no-secondary > $O/bench_rocprof.log 2>&1
python tools/timeline.py $O/prof/${TAG}_kernel_trace.csv $O/timeline.txt $O/sequence.txt > /dev/null 2>&1
rm -rf $O/prof
head -8 $O/timeline.txt | cut -c1-200
( timeout 600 python bench.py --no-secondary --no-cpu-baseline --no-roofline --stamps ) > $O/bench_stamps.log 2>&1
grep stamps $O/bench_stamps.log | cut -c1-1500
