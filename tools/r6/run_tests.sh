# usage (GPU box): bash tools/r6/run_tests.sh <tag> <pytest args>
TAG=${1:-r6g}
shift
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/$TAG
mkdir -p $O
( time timeout 2400 python -m pytest "$@" -m gpu -q -p no:cacheprovider ) > $O/pytest.log 2>&1
grep -v amdgpu.ids $O/pytest.log | tail -40 | cut -c1-400
