cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/a
mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_attn2.py tests/test_gpu_tlayer.py -m gpu -q -p no:cacheprovider -x ) > $O/pytest.log 2>&1
tail -3 $O/pytest.log | cut -c1-300
( timeout 600 python tools/r4/attn_core_bench.py ) 2>&1 | grep -v amdgpu.ids > $O/attn_core_bench.txt
grep -E "^---|attn2" $O/attn_core_bench.txt
( timeout 600 python bench.py --no-secondary --no-cpu-baseline --no-roofline ) > $O/b.log 2>&1
grep metric $O/b.log | cut -c1-200
