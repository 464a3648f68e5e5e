# round 4, first GPU call: the new B=8 oracle parity tests, the gemm tests with direct fp32 references, baseline bench + attention blocks
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4a
mkdir -p $O
( time timeout 1500 python -m pytest tests/test_gpu_b8_oracle_parity.py -q -p no:cacheprovider -x -s ) > $O/pytest_b8.log 2>&1
tail -30 $O/pytest_b8.log | cut -c1-400
( time timeout 900 python -m pytest tests/test_gpu_gemm.py -q -p no:cacheprovider -k "gemm128 or fused_stem" ) > $O/pytest_gemm.log 2>&1
tail -8 $O/pytest_gemm.log | cut -c1-300
( time timeout 900 python bench.py --no-secondary ) > $O/bench_default.log 2>&1
tail -2 $O/bench_default.log | cut -c1-3000
( timeout 600 python tools/bench_attention.py ) > $O/bench_attention.log 2>&1
tail -40 $O/bench_attention.log | cut -c1-300
