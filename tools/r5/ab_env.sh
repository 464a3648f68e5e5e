# usage (GPU box): bash tools/r5/ab_env.sh VAR   -- default bench line with TOIST_KNOBS=1 VAR=1 and VAR=0, three interleaved rounds
var=$1
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp TOIST_KNOBS=1
for i in 1 2 3; do
for v in 1 0; do
  echo "$var=$v: $(env $var=$v timeout 600 python bench.py --no-cpu-baseline --no-roofline --no-secondary 2>&1 | grep '^{"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['repeats']['ms_per_step'])")"
done; done
