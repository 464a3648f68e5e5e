# round 4: row-complete sub-layer kernels -- unit tests, model tests, A/B of the step
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4b
mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_tlayer.py -q -p no:cacheprovider -x ) > $O/pytest_tlayer.log 2>&1
tail -25 $O/pytest_tlayer.log | cut -c1-600
( time timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_b8_oracle_parity.py tests/test_gpu_captured_step.py -q -p no:cacheprovider -x ) > $O/pytest_model.log 2>&1
tail -15 $O/pytest_model.log | cut -c1-600
for i in 1 2; do
  for rows in 1 0; do
    TOIST_KNOBS=1 TOIST_ROWS=$rows timeout 600 python bench.py --no-secondary --no-cpu-baseline --no-roofline > $O/bench_rows${rows}_$i.log 2>&1
    echo "rows=$rows run $i: $(tail -1 $O/bench_rows${rows}_$i.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["repeats"]["ms_per_step"])' 2>&1 | tail -1)"
  done
done
( timeout 600 python tools/bench_attention.py ) > $O/bench_attention.log 2>&1
grep -E "us_fwd_bwd|\"ms\"" $O/bench_attention.log
