import sys, os
sys.path.insert(0, os.getcwd())
import torch
from toist_amd import kernels as k
from tools.bench_gemm import timeit
BF = torch.bfloat16
dev = torch.device("cuda")
B, H, S, dh = 8, 12, 16, 64
d = H * dh
qkv = torch.randn(B * S, 3 * d, device=dev).to(BF)
q, kk, v = (qkv[:, i * d:(i + 1) * d] for i in range(3))
bias = [torch.randn(d, device=dev) for _ in range(3)]
pad = torch.zeros(B, S, dtype=torch.uint8, device=dev)
ctx = torch.empty(B * S, d, dtype=BF, device=dev); stats = torch.empty(B * H * S * 2, device=dev)
dctx = torch.randn(B * S, d, device=dev).to(BF); dqkv = torch.empty_like(qkv)
dq, dk, dv = (dqkv[:, i * d:(i + 1) * d] for i in range(3))
k.SEED_DEV = torch.zeros(1, dtype=torch.int64, device=dev)
for p in (0.0, 0.1):
    tf = timeit(lambda: k.attn_small_fwd(q, kk, v, pad, B, H, S, dh, 0.125, p, 3, ctx, stats, *bias), 20) * 1000
    tb = timeit(lambda: k.attn_small_bwd(q, kk, v, pad, B, H, S, dh, 0.125, p, 3, stats, dctx, dq, dk, dv, *bias), 20) * 1000
    print(f"dropout {p}: fwd {tf:.1f} us  bwd {tb:.1f} us")
