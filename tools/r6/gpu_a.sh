# usage (GPU box): bash tools/r6/gpu_a.sh <tag>  -- the tests touched by the round-6 correctness fixes + a default bench line
TAG=${1:-r6a}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/$TAG
mkdir -p $O
( time timeout 1500 python -m pytest tests/test_gpu_segm.py tests/test_gpu_xdec.py tests/test_gpu_captured_step.py tests/test_gpu_criterion.py tests/test_gpu_model.py -m gpu -q -p no:cacheprovider -x --durations=8 ) > $O/pytest.log 2>&1
tail -25 $O/pytest.log | cut -c1-300
( time timeout 1200 python bench.py ) > $O/bench_default.log 2>&1
grep metric $O/bench_default.log | cut -c1-1500
