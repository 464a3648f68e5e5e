# usage (GPU box): bash tools/r4_seq.sh <tag>  -- default bench (one line) + kernel timeline and kernel SEQUENCE of the replayed step
TAG=${1:-seq}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/$TAG
mkdir -p $O
( timeout 900 python bench.py --no-secondary --no-cpu-baseline --no-roofline ) > $O/bench_plain.log 2>&1
grep metric $O/bench_plain.log | cut -c1-400
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o $TAG -- python bench.py --no-cpu-baseline --no-roofline --no-secondary > $O/bench_rocprof.log 2>&1
python tools/timeline.py $O/prof/${TAG}_kernel_trace.csv $O/timeline.txt $O/sequence.txt > /dev/null 2>&1
cp $O/prof/${TAG}_kernel_stats.csv $O/kernel_stats.csv 2>/dev/null
rm -rf $O/prof
head -8 $O/timeline.txt
