cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4k
mkdir -p $O
( timeout 600 python tools/r4/attn_core_bench.py ) > $O/attn_core_bench.log 2>&1
grep -v amdgpu.ids $O/attn_core_bench.log | grep -E "^---|attn2"
( timeout 600 python tools/bench_attention.py ) > $O/bench_attention.log 2>&1
grep -E "us_fwd_bwd|\"ms\"|launches_per" $O/bench_attention.log
( time timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider -x ) > $O/pytest.log 2>&1
tail -5 $O/pytest.log | cut -c1-300
( time timeout 900 python bench.py --no-secondary ) > $O/bench_default.log 2>&1
tail -1 $O/bench_default.log | cut -c1-600
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r4k -- python bench.py --no-cpu-baseline --no-roofline --no-secondary > $O/bench_rocprof.log 2>&1
python tools/timeline.py $O/prof/r4k_kernel_trace.csv $O/timeline.txt > /dev/null 2>&1
cp $O/prof/r4k_kernel_stats.csv $O/kernel_stats.csv 2>/dev/null
rm -rf $O/prof
head -75 $O/timeline.txt
