# usage (GPU box): bash tools/r4_final.sh <tag>  -- the round's artifact run: -m gpu suite, default bench line (all legs), kernel timeline + stats of the replayed step,
# attention blocks, attention cores, mask-head small-conv profile, glue report, stamps
TAG=${1:-final}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/$TAG
mkdir -p $O
( time timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider ) > $O/pytest.log 2>&1
tail -4 $O/pytest.log | cut -c1-300
( time timeout 1500 python bench.py ) > $O/bench_default.log 2>&1
grep metric $O/bench_default.log | cut -c1-400
( timeout 600 python bench.py --no-secondary --no-cpu-baseline --no-roofline --stamps ) > $O/bench_stamps.log 2>&1
grep stamps $O/bench_stamps.log | cut -c1-1500
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o $TAG -- python bench.py --no-cpu-baseline --no-roofline --no-secondary > $O/bench_rocprof.log 2>&1
python tools/timeline.py $O/prof/${TAG}_kernel_trace.csv $O/timeline.txt $O/sequence.txt > /dev/null 2>&1
cp $O/prof/${TAG}_kernel_stats.csv $O/kernel_stats.csv 2>/dev/null
rm -rf $O/prof
head -12 $O/timeline.txt
( timeout 600 python tools/bench_attention.py ) 2>&1 | grep -v amdgpu.ids > $O/bench_attention.json
grep -E "us_fwd_bwd|\"ms\"|launches_per" $O/bench_attention.json
( timeout 600 python tools/r4/attn_core_bench.py ) 2>&1 | grep -v amdgpu.ids > $O/attn_core_bench.txt
( timeout 600 python tools/bench_smallconv.py ) 2>&1 | grep -v amdgpu.ids > $O/smallconv.txt
tail -5 $O/smallconv.txt
( timeout 600 python bench.py --no-secondary --no-cpu-baseline --no-roofline --glue-report ) 2>&1 | grep -v amdgpu.ids | tail -80 > $O/glue.txt
head -1 $O/glue.txt
