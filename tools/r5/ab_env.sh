# usage (GPU box): bash tools/r5/ab_env.sh VAR [A B]   -- default bench line with TOIST_KNOBS=1 VAR=A (default 1) and VAR=B (default 0), three interleaved rounds
var=$1; va=${2:-1}; vb=${3:-0}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp TOIST_KNOBS=1
for i in 1 2 3; do
for v in $va $vb; do
  echo "$var=$v: $(env $var=$v timeout 600 python bench.py --no-cpu-baseline --no-roofline --no-secondary 2>&1 | grep '^{"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['repeats']['ms_per_step'])")"
done; done
