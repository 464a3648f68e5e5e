# configs[4] eager / replayed rate with and without carrying the other tail's compute copies over a step (same box, interleaved)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for i in 1 2; do for flag in True False; do
python - <<PY 2>&1 | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('carry=$flag', 'eager', d['config']['eager_pairs_per_s'], 'replay', d['config']['graph_replay_fixed_batch_pairs_per_s'])"
import sys, runpy
import toist_amd.optim as o
o.CARRY_OTHER_TAILS = $flag
sys.argv = ["bench.py", "--distill", "--batch", "4", "--no-cpu-baseline"]
try:
    runpy.run_path("bench.py", run_name="__main__")
except SystemExit:
    pass
PY
done; done
