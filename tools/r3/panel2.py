"""Round 3: panel2_kernel variants (tile word 135 | variant << 8; variant = ring stages + 16 * (32-row blocks), 1 = the round-2 kernel)
against the generic 64x64 tiles: bit-equality and us per launch (hipGraph replay of 20 launches)."""
import sys, os
sys.path.insert(0, os.getcwd())
import torch
from toist_amd import kernels as k, ops
from tools.bench_gemm import timeit
BF = torch.bfloat16
dev = torch.device("cuda")
VARS = (1, 2, 3, 4, 18, 19, 20)
shapes = ((12800, 1024, 256), (51200, 512, 128), (204800, 256, 64), (204800, 64, 256), (3328, 2048, 256), (12800, 256, 256),
          (3328, 256, 256), (800, 2048, 256), (12790, 1000, 248), (51200, 512, 256), (204800, 128, 256))
if len(sys.argv) > 1 and sys.argv[1] == "quick":
    shapes = shapes[:2] + ((12790, 1000, 248),)
for M, N, K in shapes:
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
    def plain(tile):
        return ops.linear(x, w, None, out=out, tile=tile)
    for name, f in (("fwd", fwd), ("dgrad", dgr), ("plain", plain)):
        ref = f(65).float().clone()
        row = [f"65 {timeit(lambda: f(65), 20) * 1000:6.1f}"]
        for v in VARS:
            tile = 135 | (v << 8)
            try:
                bad = 0
                for rep in range(3):
                    out.zero_()
                    got = f(tile).float()
                    bad += int((got != ref).sum())
                t = timeit(lambda: f(tile), 20) * 1000
                row.append(f"v{v} {t:6.1f}" + ("" if bad == 0 else f"(!{bad})"))
            except Exception as e:
                row.append(f"v{v}   n/a ")
        print(f"{M:7d} {N:5d} {K:4d} {name:6s} " + "  ".join(row), flush=True)
