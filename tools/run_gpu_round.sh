# usage (on the GPU box): TOIST_COMMIT=<hash> bash tools/run_gpu_round.sh <tag>
#   default bench, per-shape GEMM profile, rocprof kernel stats, timeline of the replayed step, PMC traffic passes (FETCH_SIZE and
#   WRITE_SIZE in separate runs, counters only with --kernel-trace).  Outputs under gpurun_out/<tag>/ ; copy the summaries to profiles/.
TAG=${1:-rX}
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/$TAG
mkdir -p $O
( time timeout 900 python bench.py ) > $O/bench_default.log 2>&1
tail -4 $O/bench_default.log | cut -c1-2500
timeout 600 python bench.py --static-batch --no-cpu-baseline --no-roofline > $O/bench_static_batch.log 2>&1
timeout 600 python bench.py --no-graph --no-cpu-baseline --no-roofline > $O/bench_eager.log 2>&1
timeout 600 python bench.py --masks --no-cpu-baseline --no-roofline > $O/bench_masks.log 2>&1
timeout 600 python bench.py --distill --batch 4 --no-cpu-baseline --no-roofline > $O/bench_distill.log 2>&1
timeout 600 python bench.py --profile-all --no-cpu-baseline --steps 5 --warmup 2 > $O/bench_profall.log 2>&1
mv gpurun_out/gemm_shapes.txt $O/gemm_shapes.txt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o $TAG -- python bench.py --no-cpu-baseline --no-roofline > $O/bench_rocprof.log 2>&1
python tools/timeline.py $O/prof/${TAG}_kernel_trace.csv $O/timeline.txt $O/timeline_sequence.txt
cp $O/prof/${TAG}_kernel_stats.csv $O/kernel_stats.csv
rm -rf $O/prof
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o f -- python bench.py --no-cpu-baseline --no-graph --no-roofline --steps 2 --warmup 1 > $O/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o w -- python bench.py --no-cpu-baseline --no-graph --no-roofline --steps 2 --warmup 1 > $O/pmc_write.log 2>&1
python tools/pmc_summary.py $O/pmc_fetch/f_counter_collection.csv FETCH_SIZE > $O/pmc_fetch_summary.txt 2>&1
python tools/pmc_summary.py $O/pmc_write/w_counter_collection.csv WRITE_SIZE > $O/pmc_write_summary.txt 2>&1
python tools/pmc_traffic.py $O/pmc_fetch_summary.txt $O/pmc_write_summary.txt $O/pmc_traffic.json "${TOIST_COMMIT:-unknown}"
rm -rf $O/pmc_fetch $O/pmc_write
for f in bench_static_batch bench_eager bench_masks bench_distill; do tail -1 $O/$f.log | cut -c1-260; done
du -sh $O
