cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/u
mkdir -p $O
( timeout 600 python tools/r4/rowgemm_bench.py ) 2>&1 | grep -v amdgpu.ids > $O/rowgemm_bench.txt
grep -E "rowgemm|---" $O/rowgemm_bench.txt | head -40
( timeout 1200 python -m pytest tests/test_gpu_tlayer.py tests/test_gpu_optim.py tests/test_gpu_model.py -m gpu -q -p no:cacheprovider -x -n 4 ) > $O/pytest.log 2>&1
tail -5 $O/pytest.log | cut -c1-300
run() {
  ( timeout 600 python bench.py --no-secondary --no-cpu-baseline --no-roofline --stamps "$@" ) > $O/b.log 2>&1
  echo "[$*]: $(grep metric $O/b.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["repeats"]["ms_per_step"])' 2>&1 | tail -1) $(grep stamps $O/b.log | sed 's/.*bwd.conv1.start/bwd.conv1.start/' | cut -c1-200)"
}
run
run --no-early-norm
run
run --no-early-norm
