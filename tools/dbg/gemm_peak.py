"""Ceiling of the MFMA main loop: large plain GEMMs, every tile, TFLOP/s."""
import sys, os
sys.path.insert(0, os.getcwd())
import torch
from toist_amd import kernels as k, ops
from tools.bench_gemm import timeit
BF = torch.bfloat16
dev = torch.device("cuda")
for M, N, K in ((4096, 4096, 4096), (8192, 8192, 2048), (12800, 256, 2304), (2304, 256, 12800)):
    x, w = torch.randn(M, K, device=dev).to(BF), (torch.randn(N, K, device=dev) * 0.05).to(BF)
    out = torch.empty(M, N, dtype=BF, device=dev)
    row = []
    for t in (65, 134, 130, 129, 128):
        try:
            us = timeit(lambda: ops.linear(x, w, None, out=out, tile=t), 10) * 1000
            row.append(f"{t}: {us:7.1f} us {2 * M * N * K / us / 1e6:6.0f} TF")
        except Exception as e:
            row.append(f"{t}: ERR")
    print(M, N, K, " | ".join(row), flush=True)
