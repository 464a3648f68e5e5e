# round 3: GPU test suite + default bench (with secondary legs); usage: bash tools/r3/gpu_check.sh <tag> [pytest args]
TAG=${1:-c}
shift
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/$TAG
mkdir -p $O
( time timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=10 "$@" ) > $O/pytest.log 2>&1
tail -40 $O/pytest.log | cut -c1-300
cp gpurun_out/fullsize_parity.json $O/ 2>/dev/null
( time timeout 1200 python bench.py ) > $O/bench_default.log 2>&1
tail -3 $O/bench_default.log | cut -c1-6000
