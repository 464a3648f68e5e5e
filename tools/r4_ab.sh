# usage (GPU box): bash tools/r4_ab.sh VAR v1 v2  -- A/B of one TOIST_* knob on the default bench line (two interleaved rounds, stamps)
VAR=$1; A=$2; B=$3
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/ab
for r in 1 2; do for v in $A $B; do
  ( env TOIST_KNOBS=1 $VAR=$v timeout 600 python bench.py --no-secondary --no-cpu-baseline --no-roofline --stamps ) > gpurun_out/ab/b.log 2>&1
  echo "[$VAR=$v]: $(grep metric gpurun_out/ab/b.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["repeats"]["ms_per_step"])' 2>&1 | tail -1) $(grep stamps gpurun_out/ab/b.log | sed 's/.*bwd.class_embed.start/bwd.class_embed.start/' | cut -c1-260)"
done; done
