# usage (on the GPU box): bash tools/gpu_tests.sh <tag> [pytest args]   -- the -m gpu suite + a default bench line
TAG=${1:-t}
shift
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/$TAG
mkdir -p $O
( time timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=15 "$@" ) > $O/pytest.log 2>&1
tail -60 $O/pytest.log | cut -c1-400
( time timeout 900 python bench.py ) > $O/bench_default.log 2>&1
tail -3 $O/bench_default.log | cut -c1-3000
