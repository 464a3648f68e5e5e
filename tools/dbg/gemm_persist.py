import sys, os
sys.path.insert(0, os.getcwd())
import torch
from toist_amd import kernels as k, ops
from tools.bench_gemm import timeit
BF = torch.bfloat16
dev = torch.device("cuda")
def run(M, N, K, dgrad=False):
    x, w = torch.randn(M, K, device=dev).to(BF), (torch.randn(N, K, device=dev) * 0.05).to(BF)
    if dgrad:
        dy = torch.randn(M, N, device=dev).to(BF); out = torch.empty(M, K, dtype=BF, device=dev); aux = torch.randn(M, K, device=dev).to(BF)
        return timeit(lambda: ops.linear_dgrad(dy, w, out=out, act=k.ACT_MASK_POS, aux=aux), 20) * 1000
    out = torch.empty(M, N, dtype=BF, device=dev)
    res = torch.randn(M, N, device=dev).to(BF); shift = torch.zeros(N, device=dev)
    return timeit(lambda: ops.linear(x, w, shift, out=out, res=res, act=k.ACT_RELU), 20) * 1000
shapes = [(12800, 1024, 256, 0), (12800, 256, 1024, 0), (12800, 256, 1024, 1), (12800, 1024, 256, 1), (51200, 512, 128, 0), (51200, 128, 512, 0), (204800, 256, 64, 0), (204800, 64, 256, 0), (3328, 2048, 256, 0), (3328, 256, 2048, 0), (3200, 2048, 512, 0), (3200, 512, 2048, 0)]
print(os.environ.get("TOIST_PERSIST_WGS"), " ".join(f"{run(M, N, K, bool(d)):6.1f}" for M, N, K, d in shapes), flush=True)
