cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/${1:-r5m3}; mkdir -p $O
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o masks -- python bench.py --masks $MASKS_EXTRA --no-cpu-baseline --no-roofline --no-secondary --steps 10 --warmup 3 > $O/bench_masks.log 2>&1
python tools/timeline.py $O/prof/masks_kernel_trace.csv $O/timeline.txt $O/sequence.txt > /dev/null 2>&1
cp $O/prof/masks_kernel_stats.csv $O/kernel_stats.csv; rm -rf $O/prof
grep '"metric"' $O/bench_masks.log | cut -c1-200; head -40 $O/timeline.txt | cut -c1-150
