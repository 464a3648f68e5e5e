cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/v
mkdir -p $O
run() {
  ( env TOIST_KNOBS=1 "$@" timeout 600 python bench.py --no-secondary --no-cpu-baseline --no-roofline --stamps ) > $O/b.log 2>&1
  echo "[$*]: $(grep metric $O/b.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["repeats"]["ms_per_step"])' 2>&1 | tail -1) $(grep stamps $O/b.log | sed 's/.*fwd.join/fwd.join/' | cut -c1-120)"
}
run TOIST_ROWS_FWD_MAX_M=1024
run TOIST_ROWS_FWD_MAX_M=0
run TOIST_ROWS_FWD_MAX_M=1024
run TOIST_ROWS_FWD_MAX_M=0
