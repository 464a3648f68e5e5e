cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4g
mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_attn2.py tests/test_gpu_tlayer.py -q -p no:cacheprovider ) > $O/pytest_new.log 2>&1
tail -8 $O/pytest_new.log | cut -c1-900
timeout 300 python tools/r4/nan_hunt.py 2>&1 | grep -E "^[0-9] (src|img|loss|non-finite grads)" | cut -c1-200
( time timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_b8_oracle_parity.py tests/test_gpu_captured_step.py tests/test_gpu_baseline_shapes.py tests/test_gpu_optim.py -q -p no:cacheprovider ) > $O/pytest_model.log 2>&1
tail -15 $O/pytest_model.log | cut -c1-600
for i in 1 2; do
  for cfg in "1 1 0" "1 1 1" "1 0 1" "0 0 1"; do
    set -- $cfg
    extra=""; [ "$3" = "1" ] && extra="--serial-tail"
    TOIST_KNOBS=1 TOIST_ROWS=$1 TOIST_ATTN2=$2 timeout 600 python bench.py --no-secondary --no-cpu-baseline --no-roofline $extra > $O/bench_rows$1_attn$2_serial$3_$i.log 2>&1
    echo "rows=$1 attn2=$2 serial_tail=$3 run $i: $(tail -1 $O/bench_rows$1_attn$2_serial$3_$i.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["repeats"]["ms_per_step"], d["config"]["final_loss"])' 2>&1 | tail -1)"
  done
done
( timeout 600 python tools/bench_attention.py ) > $O/bench_attention.log 2>&1
grep -E "us_fwd_bwd|\"ms\"" $O/bench_attention.log
