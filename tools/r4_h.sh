cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4h
mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_attn2.py tests/test_gpu_tlayer.py tests/test_gpu_optim.py -q -p no:cacheprovider ) > $O/pytest_new.log 2>&1
tail -8 $O/pytest_new.log | cut -c1-900
for i in 1 2; do
  for lb in 0 64 128 256; do
    TOIST_KNOBS=1 TOIST_LATE_BLOCKS=$lb timeout 600 python bench.py --no-secondary --no-cpu-baseline --no-roofline > $O/bench_late${lb}_$i.log 2>&1
    echo "late blocks=$lb run $i: $(tail -1 $O/bench_late${lb}_$i.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["repeats"]["ms_per_step"], d["config"]["final_loss"])' 2>&1 | tail -1)"
  done
  timeout 600 python bench.py --no-secondary --no-cpu-baseline --no-roofline --serial-tail > $O/bench_serial_$i.log 2>&1
  echo "serial run $i: $(tail -1 $O/bench_serial_$i.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["repeats"]["ms_per_step"], d["config"]["final_loss"])' 2>&1 | tail -1)"
done
( timeout 600 python tools/bench_attention.py ) > $O/bench_attention.log 2>&1
grep -E "us_fwd_bwd|\"ms\"|launches_per" $O/bench_attention.log
