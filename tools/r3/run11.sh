cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3m
mkdir -p $O
( timeout 1500 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_model.py tests/test_gpu_ddp.py -q -p no:cacheprovider -x 2>&1 | tail -25 ) | tee $O/pytest.log | cut -c1-300
timeout 600 python tools/bench_attention.py > $O/attention_utilisation.json 2> $O/attn.err
cat $O/attention_utilisation.json | cut -c1-1500
for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-secondary 2>&1 | grep '^{"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['repeats']['ms_per_step'])"; done
