cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3a
mkdir -p $O
timeout 900 python tools/r3/panel2.py > $O/panel2.log 2>&1
cat $O/panel2.log | cut -c1-220
for i in 1 2; do
for v in 1 3 2 19 18; do
echo "variant $v"; TOIST_PANEL_VARIANT=$v timeout 300 python bench.py --no-cpu-baseline --no-roofline 2>&1 | grep '^{"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done; done 2>&1 | tee $O/ab.log
