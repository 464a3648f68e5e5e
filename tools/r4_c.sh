cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4c
mkdir -p $O
( timeout 600 python tools/r4/rowgemm_bench.py ) > $O/rowgemm_bench.log 2>&1
cat $O/rowgemm_bench.log | grep -v amdgpu.ids
( timeout 600 python -m pytest tests/test_gpu_tlayer.py -q -p no:cacheprovider -x -k fused_layer ) > $O/pytest_tlayer.log 2>&1
tail -8 $O/pytest_tlayer.log | cut -c1-900
