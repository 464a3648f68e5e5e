import sys, os
sys.path.insert(0, os.getcwd())
import torch
from toist_amd import kernels as k, ops
from tools.bench_gemm import timeit
BF = torch.bfloat16
dev = torch.device("cuda")
def run(M, N, K, epi, tile=0):
    x, w = torch.randn(M, K, device=dev).to(BF), (torch.randn(N, K, device=dev) * 0.05).to(BF)
    out = torch.empty(M, N, dtype=BF, device=dev)
    res = torch.randn(M, N, device=dev).to(BF) if epi else None
    shift = torch.zeros(N, device=dev) if epi else None
    ms = timeit(lambda: ops.linear(x, w, shift, out=out, res=res, act=k.ACT_RELU if epi else k.ACT_NONE, tile=tile), 20)
    return ms * 1000
print("M N K | plain us | shift+res+relu us | TF plain")
for M, N in ((12800, 256), (12800, 1024), (3328, 2048), (51200, 512)):
    for K in (64, 128, 256, 512, 1024, 2048):
        if M * K > 3e8: continue
        a, b = run(M, N, K, False), run(M, N, K, True)
        print(f"{M:7d} {N:5d} {K:5d} | {a:7.1f} | {b:7.1f} | {2*M*N*K/a/1e6:6.0f}", flush=True)
x = torch.zeros(1024, device=dev).to(BF); y = torch.empty_like(x)
print("tiny add kernel us", timeit(lambda: k.add(x, x, y), 50) * 1000)
