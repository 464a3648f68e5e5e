"""Split-K with the complete epilogue (GEMM_SPLIT_EPILOGUE) on the few-tile / deep-K GEMMs (RoBERTa at 128 tokens, FFN2 of the transformer):
us per launch by slice count, forward (bias + residual, bf16 out) and data gradient; value check against split 1."""
import sys, os
sys.path.insert(0, os.getcwd())
import torch
from toist_amd import kernels as k, ops
from tools.bench_gemm import timeit
BF = torch.bfloat16
dev = torch.device("cuda")
for M, N, K in ((128, 768, 3072), (128, 768, 2304), (128, 3072, 768), (128, 768, 768), (128, 2304, 768), (3328, 256, 2048), (800, 256, 2048), (800, 256, 256),
                (800, 256, 768), (3328, 256, 768), (100, 256, 2048), (800, 1024, 256), (128, 256, 768)):
    x, w = torch.randn(M, K, device=dev).to(BF), (torch.randn(N, K, device=dev) * 0.05).to(BF)
    wt = w.t().contiguous()
    bias = torch.randn(N, device=dev); res = torch.randn(M, N, device=dev).to(BF)
    out = torch.empty(M, N, device=dev, dtype=BF)
    ref = ops.linear(x, w, bias, res=res, split_k=1).float()
    refd = ops.linear_dgrad(x, wt, res=res, split_k=1).float()
    row, rowd = [], []
    for s in (1, 2, 3, 4, 6, 8, 12, 16):
        got = ops.linear(x, w, bias, res=res, split_k=s).float()
        gotd = ops.linear_dgrad(x, wt, res=res, split_k=s).float()
        err = max(float((got - ref).abs().max()) / float(ref.abs().max()), float((gotd - refd).abs().max()) / float(refd.abs().max()))
        a = ops.linear(x, w, bias, res=res, split_k=s).float()
        same = bool((a == got).all())
        row.append(f"{s}:{timeit(lambda: ops.linear(x, w, bias, res=res, out=out, split_k=s), 20) * 1000:5.1f}" + ("" if err < 1e-2 and same else f"(!{err:.0e},{same})"))
        rowd.append(f"{s}:{timeit(lambda: ops.linear_dgrad(x, wt, res=res, out=out, split_k=s), 20) * 1000:5.1f}")
    print(f"fwd   {M:5d} {N:5d} {K:5d} auto={ops._split_for_linear(M, N, K):2d} ", " ".join(row), flush=True)
    print(f"dgrad {M:5d} {N:5d} {K:5d} auto={ops._split_for_linear(M, N, K):2d} ", " ".join(rowd), flush=True)
