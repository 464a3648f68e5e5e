"""Short-K panel kernel (tile code 135) against the generic tiles (65): us per launch (hipGraph replay) and agreement, forward
(shift + residual + ReLU) and data gradient (residual + ReLU mask), on the K <= 256 shapes of the batch-8 640^2 step."""
import sys, os
sys.path.insert(0, os.getcwd())
import torch
from toist_amd import kernels as k, ops
from tools.bench_gemm import timeit
BF = torch.bfloat16
dev = torch.device("cuda")
for M, N, K in ((12800, 1024, 256), (51200, 512, 128), (204800, 256, 64), (204800, 64, 256), (3328, 2048, 256), (12800, 256, 256), (51200, 128, 128),
                (51200, 64, 64), (3328, 256, 256), (800, 2048, 256), (12790, 1000, 248), (3200, 2048, 256), (51200, 512, 256), (204800, 128, 256)):
    x, w = torch.randn(M, K, device=dev).to(BF), (torch.randn(N, K, device=dev) * 0.05).to(BF)
    wt = w.t().contiguous()
    res = torch.randn(M, N, device=dev).to(BF); aux = torch.randn(M, N, device=dev).to(BF)
    shift = torch.randn(N, device=dev)
    out = torch.empty(M, N, dtype=BF, device=dev)
    def fwd(tile):
        return ops.linear(x, w, shift, out=out, res=res, act=k.ACT_RELU, tile=tile)
    def dgr(tile):
        k.gemm(M, N, K, k.A_ROWK, k.operand(x, K), k.B_KROW, k.operand(wt, N), out, N, res=res, ldr=N, act=k.ACT_MASK_POS, aux=aux, ldaux=N, tile=tile)
        return out
    row = []
    for name, f in (("fwd", fwd), ("dgrad", dgr)):
        ref = f(65).float().clone()
        got = f(135).float().clone()
        err = float((got - ref).abs().max())
        t65, t135, t0 = (timeit(lambda: f(t), 20) * 1000 for t in (65, 135, 0))
        row.append(f"{name}: 65 {t65:6.1f}  panel {t135:6.1f}  auto {t0:6.1f}" + ("" if err == 0 else f" (!{err:.2e})"))
    print(f"{M:7d} {N:5d} {K:4d}  " + "   ".join(row), flush=True)

import sys; sys.exit(0)
M, N, K = 12800, 1024, 256
x, w = torch.randn(M, K, device=dev).to(BF), (torch.randn(N, K, device=dev) * 0.05).to(BF)
wt = w.t().contiguous()
res = torch.randn(M, N, device=dev).to(BF); aux = torch.randn(M, N, device=dev).to(BF)
shift = torch.randn(N, device=dev); out = torch.empty(M, N, dtype=BF, device=dev)
for tile in (135, 65):
    for name in ("fwd", "dgrad", "fwd-plain"):
        row = []
        for fl in (0, 256, 512, 768):
            if name == "fwd":
                f = lambda: ops.linear(x, w, shift, out=out, res=res, act=k.ACT_RELU, tile=tile, flags=fl)
            elif name == "fwd-plain":
                f = lambda: ops.linear(x, w, None, out=out, tile=tile, flags=fl)
            else:
                f = lambda: k.gemm(M, N, K, k.A_ROWK, k.operand(x, K), k.B_KROW, k.operand(wt, N), out, N, res=res, ldr=N, act=k.ACT_MASK_POS, aux=aux, ldaux=N, tile=tile, flags=fl)
            row.append(f"{timeit(f, 20) * 1000:6.1f}")
        print(tile, name, " | ".join(row), flush=True)

