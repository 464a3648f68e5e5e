# usage (GPU box): bash tools/r4_t.sh <tag>  -- the -m gpu suite (stop at first failure), a plain bench line with stamps, the glue report
TAG=${1:-t}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/$TAG
mkdir -p $O
( time timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider -x -n 4 ) > $O/pytest.log 2>&1 || ( time timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider -x ) > $O/pytest.log 2>&1
tail -25 $O/pytest.log | cut -c1-300
( timeout 600 python bench.py --no-secondary --no-cpu-baseline --no-roofline --stamps ) > $O/bench_plain.log 2>&1
grep -E "metric|stamps" $O/bench_plain.log | cut -c1-1200
( timeout 600 python bench.py --no-secondary --no-cpu-baseline --no-roofline --glue-report ) 2>&1 | grep -v amdgpu.ids | tail -100 > $O/glue.txt
head -1 $O/glue.txt
