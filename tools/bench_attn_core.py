"""Attention core forward: fused kernel (csrc/attn.hip) vs score GEMM + softmax + context GEMM.  GPU only."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from toist_amd import kernels as k, ops  # noqa: E402
from tools.bench_gemm import timeit  # noqa: E402

BF = torch.bfloat16
dev = torch.device("cuda")
B, H, dh = 8, 8, 32
d = H * dh
k.SEED_DEV = torch.zeros(1, dtype=torch.int64, device=dev)
for Sq, Sk in [(416, 416), (100, 416), (100, 100)]:
    q = torch.randn(B * Sq, d, device=dev).to(BF)
    kk = torch.randn(B * Sk, d, device=dev).to(BF)
    v = torch.randn(B * Sk, d, device=dev).to(BF)
    pad = torch.zeros(B, Sk, dtype=torch.uint8, device=dev)
    ld = ops.round8(Sk)
    s = torch.empty(B * H, Sq, ld, dtype=BF, device=dev)
    p0, pu0 = torch.empty_like(s), torch.empty_like(s)
    c = torch.empty(B * Sq, d, dtype=BF, device=dev)
    sc = 1 / math.sqrt(dh)

    def unfused():
        ops.attn_scores(q, kk, B, H, Sq, Sk, dh, sc, out=s)
        k.softmax_fwd(s, pad, B, H, Sq, Sk, ld, p0, pu0, 0.1, 7)
        ops.attn_context(pu0, v, B, H, Sq, Sk, dh, c)

    fused = lambda: k.attn_fwd(q, kk, v, pad, B, H, Sq, Sk, dh, sc, p0, pu0, 0.1, 7, c)
    print(f"Sq={Sq} Sk={Sk}: three kernels {1000 * timeit(unfused, 30):7.1f} us   fused {1000 * timeit(fused, 30):7.1f} us", flush=True)
    # backward: fused kernel vs dV / dP GEMMs + softmax backward + dQ / dK GEMMs
    fused()
    dctx = torch.randn(B * Sq, d, device=dev).to(BF)
    dq, dk, dv = torch.empty(B * Sq, d, dtype=BF, device=dev), torch.empty(B * Sk, d, dtype=BF, device=dev), torch.empty(B * Sk, d, dtype=BF, device=dev)

    def sm_bwd(dp):
        ds = torch.empty_like(dp)
        k.softmax_bwd(p0, dp, B * H * Sq, Sk, ld, ds, 0.1, 7)
        return ds
    unfused_b = lambda: ops.attn_backward(pu0, sc, q, kk, v, dctx, B, H, Sq, Sk, dh, dq, dk, dv, sm_bwd)
    fused_b1 = lambda: k.attn_bwd(q, kk, v, p0, pu0, c, dctx, B, H, Sq, Sk, dh, sc, 0.1, dq, dk, dv, variant=1)
    fused_b2 = lambda: k.attn_bwd(q, kk, v, p0, pu0, c, dctx, B, H, Sq, Sk, dh, sc, 0.1, dq, dk, dv, variant=2)
    fused_b4 = lambda: k.attn_bwd(q, kk, v, p0, pu0, c, dctx, B, H, Sq, Sk, dh, sc, 0.1, dq, dk, dv, variant=2, q_splits=4)
    fused_b2s = lambda: k.attn_bwd(q, kk, v, p0, pu0, c, dctx, B, H, Sq, Sk, dh, sc, 0.1, dq, dk, dv, variant=2, q_splits=2)
    print(f"Sq={Sq} Sk={Sk}: backward five kernels {1000 * timeit(unfused_b, 30):7.1f} us   fused key-major {1000 * timeit(fused_b1, 30):7.1f} us   "
          f"fused query-major {1000 * timeit(fused_b2, 30):7.1f} us   x2 splits {1000 * timeit(fused_b2s, 30):7.1f} us   x4 splits "
          f"{1000 * timeit(fused_b4, 30):7.1f} us", flush=True)
