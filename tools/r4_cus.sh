cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/cus
mkdir -p $O
run() {
  ( timeout 600 python bench.py --no-secondary --no-cpu-baseline --no-roofline "$@" ) > $O/b.log 2>&1
  echo "[$*]: $(grep metric $O/b.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["repeats"]["ms_per_step"])' 2>&1 | tail -1) $(grep stamps $O/b.log | cut -c60-900)"
}
run
run --split-graph --stamps
run --split-graph --text-cus 64 --stamps
run --split-graph --text-cus 32 --stamps
run --split-graph --text-cus 16 --stamps
run --split-graph
run --split-graph --text-cus 32
