# round 3 artifacts (on the GPU box): bash tools/r3/gpu_round.sh <tag>
#   full -m gpu suite, default bench (with secondary legs + live PMC), eager / static-batch legs, per-shape GEMM profile, rocprof kernel
#   stats + timeline of the replayed step, attention block benchmark.  Outputs under gpurun_out/<tag>/ ; copy the summaries to profiles/.
TAG=${1:-r03}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/$TAG
mkdir -p $O
( time timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=10 ) > $O/pytest.log 2>&1
tail -25 $O/pytest.log | cut -c1-250
cp gpurun_out/fullsize_parity.json $O/ 2>/dev/null
( time timeout 600 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1; tail -4 $O/smoke.log | cut -c1-200
( time timeout 1500 python bench.py ) > $O/bench_default.log 2>&1
grep '^{"metric"' $O/bench_default.log | tail -1 > $O/bench_default.json
python - <<PY
import json
d = json.load(open("$O/bench_default.json"))
print(d["value"], d["ms_per_step"], d.get("repeats"), d["roofline"]["frac"], d["roofline"]["traffic"], d["roofline"]["avg_launch_us"])
for k_, v in d.get("secondary", {}).items(): print(k_, v.get("value"), v.get("error"))
PY
timeout 600 python bench.py --static-batch --no-cpu-baseline --no-roofline --no-secondary 2>&1 | grep '^{"metric"' > $O/bench_static_batch.json
timeout 600 python bench.py --no-graph --no-cpu-baseline --no-roofline --no-secondary 2>&1 | grep '^{"metric"' > $O/bench_eager.json
timeout 600 python bench.py --profile-all --no-cpu-baseline --no-secondary --steps 5 --warmup 2 > $O/bench_profall.log 2>&1
mv gpurun_out/gemm_shapes.txt $O/gemm_shapes_eager.txt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o $TAG -- python bench.py --no-cpu-baseline --no-roofline --no-secondary > $O/bench_rocprof.log 2>&1
python tools/timeline.py $O/prof/${TAG}_kernel_trace.csv $O/timeline_graph_step.txt $O/timeline_graph_step_kernel_sequence.txt
cp $O/prof/${TAG}_kernel_stats.csv $O/bench_kernel_stats.csv
rm -rf $O/prof
head -30 $O/timeline_graph_step.txt | cut -c1-140
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o f -- python bench.py --no-cpu-baseline --no-graph --no-roofline --no-secondary --steps 2 --warmup 1 > $O/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o w -- python bench.py --no-cpu-baseline --no-graph --no-roofline --no-secondary --steps 2 --warmup 1 > $O/pmc_write.log 2>&1
python tools/pmc_summary.py $O/pmc_fetch/f_counter_collection.csv FETCH_SIZE > $O/pmc_fetch_summary.txt 2>&1
python tools/pmc_summary.py $O/pmc_write/w_counter_collection.csv WRITE_SIZE > $O/pmc_write_summary.txt 2>&1
python tools/pmc_traffic.py $O/pmc_fetch_summary.txt $O/pmc_write_summary.txt $O/pmc_traffic.json "${TOIST_COMMIT:-unknown}"
rm -rf $O/pmc_fetch $O/pmc_write
timeout 600 python tools/bench_attention.py > $O/attention_utilisation.json 2> $O/attn.err
tail -8 $O/attention_utilisation.json
du -sh $O
