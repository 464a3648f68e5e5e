cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4d
mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_attn2.py -q -p no:cacheprovider ) > $O/pytest_attn2.log 2>&1
tail -30 $O/pytest_attn2.log | cut -c1-500
( timeout 600 python -m pytest tests/test_gpu_tlayer.py -q -p no:cacheprovider ) > $O/pytest_tlayer.log 2>&1
tail -8 $O/pytest_tlayer.log | cut -c1-900
( timeout 600 python tools/r4/rowgemm_bench.py ) > $O/rowgemm_bench.log 2>&1
grep -v amdgpu.ids $O/rowgemm_bench.log | grep -E "^---|fused"
