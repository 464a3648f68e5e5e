# round 3: every dispatcher switch introduced this round against the default, same box, two interleaved rounds (KNOBS=1 build)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for i in 1 2; do
for e in "X=0" "TOIST_PANEL_VARIANT=1" "TOIST_GEMM128=0" "TOIST_GEMM128W=0" "TOIST_GEMM256W=0" "TOIST_G8_RING=4 TOIST_G8W_RING=4" "TOIST_FILL_OUTSIDE_GRAPH=0"; do
env $e timeout 600 python bench.py --no-cpu-baseline --no-roofline --no-secondary --repeats 3 2>/dev/null | grep '^{"metric"' | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('%-36s %7.1f img/s  ms/step %s' % ('$e', d['value'], d['repeats']['ms_per_step']))"
done; done
