# usage (GPU box): bash tools/r6_final.sh <tag>  -- the round's artifact run: -m gpu suite, smoke, default bench line (all legs, kernel-trace roofline), stamps, kernel timeline + stats of
# the replayed step, attention blocks + XCD-resident decoder phases (profiles/r06_attention_utilisation.json), LDS conflict counters, configs[2] / configs[4] timelines, LSAP timing, gemm128 model
TAG=${1:-final6}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/$TAG
mkdir -p $O
( time timeout 2700 python -m pytest tests -m gpu -q -p no:cacheprovider ) > $O/pytest.log 2>&1
grep -v amdgpu.ids $O/pytest.log | tail -4 | cut -c1-300
( timeout 600 python -c 'import __graft_entry__ as g; g.smoke()' ) 2>&1 | grep -v amdgpu.ids | tail -2 | tee $O/smoke.txt
( time timeout 1500 python bench.py ) > $O/bench_default.log 2>&1
grep metric $O/bench_default.log | cut -c1-300
cp gpurun_out/bench_kernel_stats.csv $O/bench_kernel_stats_from_bench.csv 2>/dev/null
( timeout 600 python bench.py --no-secondary --no-cpu-baseline --no-roofline --stamps ) > $O/bench_stamps.log 2>&1
grep stamps $O/bench_stamps.log | cut -c1-1500
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o $TAG -- python bench.py --no-cpu-baseline --no-roofline --no-secondary > $O/bench_rocprof.log 2>&1
python tools/timeline.py $O/prof/${TAG}_kernel_trace.csv $O/timeline.txt $O/sequence.txt > /dev/null 2>&1
cp $O/prof/${TAG}_kernel_stats.csv $O/kernel_stats.csv 2>/dev/null
rm -rf $O/prof
head -12 $O/timeline.txt
( timeout 600 python tools/bench_attention.py ) 2>&1 | grep -v amdgpu.ids > $O/attention.txt
( timeout 300 python tools/r5/xdec_bench.py --train --bwd ) 2>&1 | grep -v amdgpu.ids > $O/xdec_bench.txt
python tools/r6/attention_util.py $O/attention.txt $O/xdec_bench.txt $O/attention_utilisation.json
( bash tools/pmc_lds.sh ) > $O/pmc_lds.txt 2>&1
( timeout 600 python bench.py --no-secondary --no-cpu-baseline --no-roofline --glue-report ) 2>&1 | grep -v amdgpu.ids | tail -80 > $O/glue.txt
head -1 $O/glue.txt
# configs[2]: kernel timeline of the replayed step
( bash tools/r5/masks_prof.sh $TAG/masks ) > $O/masks_prof.txt 2>&1
# configs[4]: the any-batch replayed step
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/profd -o d -- python bench.py --distill --batch 4 --no-cpu-baseline --no-roofline > $O/bench_distill_rocprof.log 2>&1
python tools/timeline.py $O/profd/d_kernel_trace.csv $O/distill_timeline.txt $O/distill_sequence.txt > /dev/null 2>&1
rm -rf $O/profd
head -8 $O/distill_timeline.txt
( timeout 300 python tools/r6/lsap_bench.py ) 2>&1 | grep -v amdgpu.ids > $O/lsap_bench.txt
( timeout 300 python tools/r6/gemm128_model.py ) 2>&1 | grep -v amdgpu.ids > $O/gemm128_model.txt
tail -1 $O/gemm128_model.txt
