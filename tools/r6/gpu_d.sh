# usage (GPU box): bash tools/r6/gpu_d.sh <tag>  -- general-size mask head tests, DDP structure tests, the gemm128 stream-K model probe
TAG=${1:-r6d}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/$TAG
mkdir -p $O
( time timeout 1500 python -m pytest tests/test_gpu_segm.py tests/test_gpu_model.py tests/test_gpu_captured_step.py tests/test_gpu_ddp.py tests/test_gpu_fullsize_parity.py tests/test_gpu_b8_masks_parity.py -m gpu -q -p no:cacheprovider -s --durations=6 ) > $O/pytest.log 2>&1
grep -v amdgpu.ids $O/pytest.log | tail -25 | cut -c1-400
( timeout 600 python tools/r6/gemm128_model.py ) 2>&1 | grep -v amdgpu.ids | tee $O/gemm128_model.txt
