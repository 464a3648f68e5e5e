cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/m
mkdir -p $O
( timeout 900 python bench.py --masks --no-secondary --no-cpu-baseline --no-roofline ) > $O/bench_masks.log 2>&1
grep metric $O/bench_masks.log | cut -c1-300
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o m -- python bench.py --masks --no-cpu-baseline --no-roofline --no-secondary > $O/bench_rocprof.log 2>&1
python tools/timeline.py $O/prof/m_kernel_trace.csv $O/timeline.txt $O/sequence.txt > /dev/null 2>&1
rm -rf $O/prof
head -45 $O/timeline.txt
