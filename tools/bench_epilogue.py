"""Cost of the fused epilogue variants on the layer3 1x1-conv shape (GPU only)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from toist_amd import kernels as k, ops  # noqa: E402
from tools.bench_gemm import timeit  # noqa: E402

BF = torch.bfloat16
dev = torch.device("cuda")
for M, N, K in [(12800, 1024, 256), (12800, 256, 1024), (51200, 512, 128)]:
    x = torch.randn(M, K, device=dev).to(BF)
    w = (torch.randn(N, K, device=dev) * 0.05).to(BF)
    res = torch.randn(M, N, device=dev).to(BF)
    shift = torch.randn(N, device=dev)
    out = torch.empty(M, N, dtype=BF, device=dev)
    fl = 2 * M * N * K
    rows = [("plain", lambda: ops.linear(x, w, out=out)),
            ("+bias", lambda: ops.linear(x, w, shift, out=out)),
            ("+bias+relu", lambda: ops.linear(x, w, shift, act=k.ACT_RELU, out=out)),
            ("+bias+res+relu", lambda: ops.linear(x, w, shift, act=k.ACT_RELU, res=res, out=out)),
            ("dgrad +res +mask(aux)", lambda: ops.linear_dgrad(x, w.t().contiguous()[:K] if False else torch.empty(K, N, device=dev).to(BF), out=out, res=res, act=k.ACT_MASK_POS, aux=res))]
    for name, fn in rows:
        ms = timeit(fn, 30)
        print(f"M{M} N{N} K{K} {name:24s} {1000 * ms:7.1f} us  {fl / ms / 1e9:6.0f} TFLOP/s", flush=True)
