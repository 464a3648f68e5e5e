import sys, os
sys.path.insert(0, os.getcwd())
import torch
from toist_amd import kernels as k, ops
from tools.bench_gemm import timeit
BF = torch.bfloat16
dev = torch.device("cuda")
def run(M, N, K, flags, epi=True):
    x, w = torch.randn(M, K, device=dev).to(BF), (torch.randn(N, K, device=dev) * 0.05).to(BF)
    out = torch.empty(M, N, dtype=BF, device=dev)
    res = torch.randn(M, N, device=dev).to(BF) if epi else None
    shift = torch.zeros(N, device=dev) if epi else None
    return timeit(lambda: ops.linear(x, w, shift, out=out, res=res, act=k.ACT_RELU if epi else k.ACT_NONE, flags=flags), 20) * 1000
print("M N K | full | no k-loop | no epilogue | neither   (us, shift+res+relu epilogue)")
for M, N, K in ((12800, 1024, 256), (12800, 256, 1024), (12800, 256, 256), (51200, 512, 128), (204800, 256, 64), (3328, 2048, 256), (3328, 256, 2048), (800, 256, 256), (128, 768, 768), (128, 3072, 768)):
    print(f"{M:7d} {N:5d} {K:5d} | {run(M,N,K,0):6.1f} | {run(M,N,K,256):6.1f} | {run(M,N,K,512):6.1f} | {run(M,N,K,768):6.1f}", flush=True)
