# usage (on the GPU box): bash tools/gpu_round.sh <tag>   -- -m gpu suite, default bench, kernel timeline of the replayed step
TAG=${1:-t}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/$TAG
mkdir -p $O
( time timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider -x ) > $O/pytest.log 2>&1
tail -5 $O/pytest.log | cut -c1-300
( time timeout 900 python bench.py ) > $O/bench_default.log 2>&1
tail -3 $O/bench_default.log | cut -c1-1200
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o $TAG -- python bench.py --no-cpu-baseline --no-roofline > $O/bench_rocprof.log 2>&1
python tools/timeline.py $O/prof/${TAG}_kernel_trace.csv $O/timeline.txt > /dev/null 2>&1
cp $O/prof/${TAG}_kernel_stats.csv $O/kernel_stats.csv 2>/dev/null
rm -rf $O/prof
head -70 $O/timeline.txt
