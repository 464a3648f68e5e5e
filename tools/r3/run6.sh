cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3o
mkdir -p $O
timeout 900 python tools/r3/panel2.py > $O/panel3.log 2>&1
cat $O/panel3.log | cut -c1-220
timeout 600 python tools/r3/panel_phases.py > $O/phases.log 2>&1
grep "v 2\|v18" $O/phases.log | cut -c1-220
for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-secondary 2>&1 | grep '^{"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['repeats']['ms_per_step'])"; done
