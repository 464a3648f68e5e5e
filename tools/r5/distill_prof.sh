# configs[4] (bench.py --distill --batch 4): kernel timeline of the replayed fixed-batch step
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/${1:-r5d1}; mkdir -p $O
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o dist -- python bench.py --distill --batch 4 --no-cpu-baseline --steps 6 --warmup 3 > $O/bench_distill.log 2>&1
python tools/timeline.py $O/prof/dist_kernel_trace.csv $O/timeline.txt $O/sequence.txt > /dev/null 2>&1
cp $O/prof/dist_kernel_stats.csv $O/kernel_stats.csv; rm -rf $O/prof
grep '"metric"' $O/bench_distill.log | cut -c1-300; head -45 $O/timeline.txt | cut -c1-150
