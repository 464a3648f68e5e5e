cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3n
mkdir -p $O
export TOIST_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $O/b2.log 2>&1
grep '^{"metric"' $O/b2.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config'].get('parameters_identical_across_ranks'), json.dumps(d['config'].get('parameters_differing'))[:1500])"
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-graph > $O/b2e.log 2>&1
grep '^{"metric"' $O/b2e.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('eager', d['config'].get('parameters_identical_across_ranks'), json.dumps(d['config'].get('parameters_differing'))[:1500])"
tail -5 $O/b2e.log | cut -c1-300
