"""Latency-bound linears (few workgroups, long reduction): the 64x64x64 tile with its 3-slot DMA ring against an 8-slot ring -- an experiment that is not in the tree
(profiles/r05_deep_ring_skinny_gemm.txt; without it both columns run the shipped kernel) (csrc/gemm.hip launch_variant: at most one workgroup per CU and >= 5 k-tiles per slice).  hipGraph-replayed launches, us per call,
and the result checked against an fp32 matmul.  GPU only.  Usage: python tools/r5/skinny_bench.py [--iters 50]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from toist_amd import kernels as k, ops  # noqa: E402
from tools.bench_gemm import timeit  # noqa: E402

BF = torch.bfloat16


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=50)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    shapes = [  # (M, N, K, what)
        (128, 768, 768, "RoBERTa attention output"), (128, 2304, 768, "RoBERTa packed q|k|v"), (128, 3072, 768, "RoBERTa FFN1"),
        (128, 768, 3072, "RoBERTa FFN2"), (128, 768, 2304, "RoBERTa d(q|k|v) -> dx"), (3328, 256, 2048, "encoder FFN2"),
        (800, 256, 2048, "decoder-size FFN2"), (3328, 256, 768, "text resizer-like"), (800, 256, 256, "heads"), (3328, 2048, 256, "encoder FFN1 (K = 256: not eligible)"),
    ]
    print("# M N K | forward (w row-major [N, K]): old ring us, deep ring us (split) | data gradient (w k-major): old, deep | max rel err deep vs fp32")
    for M, N, K, what in shapes:
        x = torch.randn(M, K, device=dev).to(BF)
        w = (torch.randn(N, K, device=dev) / K ** 0.5).to(BF)
        wt = (torch.randn(K, N, device=dev) / K ** 0.5).to(BF)      # dgrad: dy [M, K'] @ w [K', N']: reuse names: dx[M, N] = x[M, K] @ wt[K, N]
        b = torch.randn(N, device=dev)
        ref = x.float() @ w.float().t() + b
        reft = x.float() @ wt.float()
        sp = ops._split_for_linear(M, N, K)
        row = []
        for split in sorted({1, sp, max(1, sp // 2)}):
            o_old = torch.empty(M, N, dtype=BF, device=dev)
            o_new = torch.empty(M, N, dtype=BF, device=dev)
            t_old = timeit(lambda: ops.linear(x, w, b, out=o_old, tile=65 | (3 << 8), split_k=split), a.iters) * 1e3
            t_new = timeit(lambda: ops.linear(x, w, b, out=o_new, split_k=split), a.iters) * 1e3
            err = ((o_new.float() - ref).abs().max() / ref.abs().max()).item()
            same = torch.equal(o_old, o_new)
            row.append(f"split {split}: {t_old:6.1f} -> {t_new:6.1f} us (err {err:.1e}{'' if same else ', differs from old ring'})")
        print(f"{M:5d} {N:5d} {K:5d} fwd   " + " | ".join(row) + f"   [{what}]")
        row = []
        for split in sorted({1, sp, max(1, sp // 2)}):
            d_new = torch.empty(M, N, dtype=BF, device=dev)
            t_new = timeit(lambda: ops.linear_dgrad(x, wt, out=d_new, split_k=split), a.iters) * 1e3
            k.FORCE_TILE = 65 | (3 << 8)
            d_old = torch.empty(M, N, dtype=BF, device=dev)
            t_old = timeit(lambda: ops.linear_dgrad(x, wt, out=d_old, split_k=split), a.iters) * 1e3
            k.FORCE_TILE = 0
            err = ((d_new.float() - reft).abs().max() / reft.abs().max()).item()
            row.append(f"split {split}: {t_old:6.1f} -> {t_new:6.1f} us (err {err:.1e}{'' if torch.equal(d_old, d_new) else ', differs from old ring'})")
        print(f"{M:5d} {N:5d} {K:5d} dgrad " + " | ".join(row))


if __name__ == "__main__":
    main()
