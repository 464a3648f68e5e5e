# usage (GPU box): bash tools/r5_final.sh <tag>  -- the round's artifact run: -m gpu suite, default bench line (all legs), stamps, kernel timeline + stats of the replayed step,
# the XCD-resident decoder launches' phase profiles, LDS conflict counters, XCD barrier probe
TAG=${1:-final5}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/$TAG
mkdir -p $O
( time timeout 2700 python -m pytest tests -m gpu -q -p no:cacheprovider ) > $O/pytest.log 2>&1
tail -4 $O/pytest.log | cut -c1-300
( timeout 600 python -c 'import __graft_entry__ as g; g.smoke()' ) 2>&1 | grep -v amdgpu.ids | tail -2 | tee $O/smoke.txt
( time timeout 1500 python bench.py ) > $O/bench_default.log 2>&1
grep metric $O/bench_default.log | cut -c1-400
( timeout 600 python bench.py --no-secondary --no-cpu-baseline --no-roofline --stamps ) > $O/bench_stamps.log 2>&1
grep stamps $O/bench_stamps.log | cut -c1-1500
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o $TAG -- python bench.py --no-cpu-baseline --no-roofline --no-secondary > $O/bench_rocprof.log 2>&1
python tools/timeline.py $O/prof/${TAG}_kernel_trace.csv $O/timeline.txt $O/sequence.txt > /dev/null 2>&1
cp $O/prof/${TAG}_kernel_stats.csv $O/kernel_stats.csv 2>/dev/null
rm -rf $O/prof
head -14 $O/timeline.txt
( timeout 300 python tools/r5/xdec_bench.py --train --bwd ) 2>&1 | grep -v amdgpu.ids > $O/xdec_bench.txt
grep -E "decoder forward|backward launch|one-launch" $O/xdec_bench.txt
( bash tools/pmc_lds.sh ) > $O/pmc_lds.txt 2>&1
( timeout 120 build/xcd_barrier_probe ) > $O/xcd_barrier.txt 2>&1
( timeout 600 python bench.py --no-secondary --no-cpu-baseline --no-roofline --glue-report ) 2>&1 | grep -v amdgpu.ids | tail -80 > $O/glue.txt
head -1 $O/glue.txt
# configs[2]: kernel timeline of the replayed step, the fused mask stages against their HBM bytes
( bash tools/r5/masks_prof.sh $TAG/masks ) > $O/masks_prof.txt 2>&1
( timeout 300 python tools/r5/maskstage_bench.py ) 2>&1 | grep -v amdgpu.ids > $O/maskstage_bench.txt
cat $O/maskstage_bench.txt
