cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q -k "gemm128 or criterion or contrastive" 2>&1 | tail -3
timeout 800 python tools/r3/gemm128.py 2>&1 | grep -v amdgpu.ids | cut -c1-200
timeout 800 python tools/r3/conv128.py 2>&1 | grep -v amdgpu.ids | head -8 | cut -c1-200
