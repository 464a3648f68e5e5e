cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3k
mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_captured_step.py tests/test_gpu_fullsize_parity.py -q -p no:cacheprovider -x 2>&1 | tail -30 ) | tee $O/pytest.log | cut -c1-300
timeout 600 python bench.py --mixed-sizes --steps 12 --no-cpu-baseline --no-roofline 2>&1 | tail -3 | cut -c1-1200
