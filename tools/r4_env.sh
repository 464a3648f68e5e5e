# runtime knobs of the hipGraph executor against the default bench line (plain runs, no secondary legs)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/envs
mkdir -p $O
run() {
  ( env "$@" timeout 600 python bench.py --no-secondary --no-cpu-baseline --no-roofline ) > $O/b.log 2>&1
  echo "[$*]: $(grep metric $O/b.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["repeats"]["ms_per_step"])' 2>&1 | tail -1)"
}
run A=1
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run DEBUG_HIP_GRAPH_BATCH_SIZE=1
run DEBUG_HIP_GRAPH_BATCH_SIZE=8
run DEBUG_HIP_GRAPH_BATCH_SIZE=64
run DEBUG_HIP_GRAPH_BATCH_SIZE=1024
run DEBUG_HIP_FORCE_GRAPH_QUEUES=2
run DEBUG_HIP_FORCE_GRAPH_QUEUES=8
run GPU_MAX_HW_QUEUES=8
run DEBUG_CLR_MAX_BATCH_SIZE=1
run A=2
