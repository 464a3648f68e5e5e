cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4e
mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_tlayer.py tests/test_gpu_attn2.py -q -p no:cacheprovider -x ) > $O/pytest_new.log 2>&1
tail -8 $O/pytest_new.log | cut -c1-900
( time timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_b8_oracle_parity.py tests/test_gpu_captured_step.py tests/test_gpu_baseline_shapes.py -q -p no:cacheprovider -x ) > $O/pytest_model.log 2>&1
tail -15 $O/pytest_model.log | cut -c1-600
for i in 1 2; do
  for cfg in "1 1" "1 0" "0 0"; do
    set -- $cfg
    TOIST_KNOBS=1 TOIST_ROWS=$1 TOIST_ATTN2=$2 timeout 600 python bench.py --no-secondary --no-cpu-baseline --no-roofline > $O/bench_rows$1_attn$2_$i.log 2>&1
    echo "rows=$1 attn2=$2 run $i: $(tail -1 $O/bench_rows$1_attn$2_$i.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["repeats"]["ms_per_step"])' 2>&1 | tail -1)"
  done
done
