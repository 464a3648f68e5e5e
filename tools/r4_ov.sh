# overlap experiments: default | --no-overlap | --split-graph (plain bench lines), + kernel sequence of the split-graph step
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/ov
mkdir -p $O
for v in "" "--no-overlap" "--split-graph"; do
  ( timeout 600 python bench.py --no-secondary --no-cpu-baseline --no-roofline $v ) > $O/b.log 2>&1
  echo "variant [$v]: $(grep metric $O/b.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["repeats"]["ms_per_step"])')"
done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o ov -- python bench.py --no-cpu-baseline --no-roofline --no-secondary --split-graph > $O/bench_rocprof.log 2>&1
python tools/timeline.py $O/prof/ov_kernel_trace.csv $O/timeline_split.txt $O/sequence_split.txt > /dev/null 2>&1
rm -rf $O/prof
head -3 $O/timeline_split.txt; tail -8 $O/timeline_split.txt
