# usage (GPU box): bash tools/r5/masks_ab.sh  -- configs[2] bench line (every step a different batch) and the frozen recipe
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for extra in "" "--frozen"; do
  echo "masks $extra: $(timeout 600 python bench.py --masks $extra --no-cpu-baseline --no-roofline --no-secondary --steps 10 --warmup 3 2>&1 | grep '^{"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")"
done
