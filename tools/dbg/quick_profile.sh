# usage (GPU box): bash tools/dbg/quick_profile.sh <tag> [bench flags] -- kernel stats + timeline of the replayed step
TAG=${1:-q}
shift
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/$TAG
mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o $TAG -- python bench.py --no-cpu-baseline --no-roofline "$@" > $O/bench_rocprof.log 2>&1
python tools/timeline.py $O/prof/${TAG}_kernel_trace.csv $O/timeline.txt $O/timeline_sequence.txt
cp $O/prof/${TAG}_kernel_stats.csv $O/kernel_stats.csv
rm -rf $O/prof
head -45 $O/timeline.txt
