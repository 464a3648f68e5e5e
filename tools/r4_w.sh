cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/w
mkdir -p $O
run() {
  ( env TOIST_KNOBS=1 "$@" timeout 600 python bench.py --no-secondary --no-cpu-baseline --no-roofline --stamps ) > $O/b.log 2>&1
  echo "[$*]: $(grep metric $O/b.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["repeats"]["ms_per_step"], d["config"].get("final_loss"))' 2>&1 | tail -1) $(grep stamps $O/b.log | sed 's/.*bwd.class_embed.start/bwd.class_embed.start/' | cut -c1-330)"
}
run TOIST_FORK_WGRADS=1
run TOIST_FORK_WGRADS=0
run TOIST_FORK_WGRADS=1
run TOIST_FORK_WGRADS=0
( time timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider -x -n 4 ) > $O/pytest.log 2>&1
tail -5 $O/pytest.log | cut -c1-300
