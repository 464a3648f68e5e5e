# usage (GPU box): bash tools/r5/ab_lib.sh OTHER.so  -- default bench line with the in-tree library and with OTHER.so (TOIST_HIP_LIB), two interleaved rounds
other=$1
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for i in 1 2; do
for lib in "" "$other"; do
  echo "lib=${lib:-in-tree}: $(TOIST_HIP_LIB=$lib timeout 600 python bench.py --no-cpu-baseline --no-roofline --no-secondary 2>&1 | grep '^{"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['repeats']['ms_per_step'])")"
done; done
