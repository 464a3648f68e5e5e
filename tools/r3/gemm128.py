"""Round 3: gemm128_kernel (tile code 136: 128 x 128 tiles, 64 x 64 wave tiles, k-halves on wave pairs) against the 64 x 64 (65) and
128 x 64 (130) generic tiles on the deep row-major GEMMs of the step: us per launch (hipGraph replay of 20 launches) and max deviation
(the k-half split changes the summation order: values agree to a bf16 ulp, not bit for bit)."""
import sys, os
sys.path.insert(0, os.getcwd())
import torch
from toist_amd import kernels as k, ops
from tools.bench_gemm import timeit
BF = torch.bfloat16
dev = torch.device("cuda")
shapes = ((12800, 256, 1024), (12800, 512, 1024), (12800, 1024, 512), (51200, 128, 512), (51200, 256, 512), (3200, 512, 2048), (3200, 2048, 1024), (3200, 1024, 2048),
          (3328, 256, 2048), (3328, 2048, 256), (12790, 250 // 8 * 8, 768), (4096, 4096, 4096))
for M, N, K in shapes:
    x, w = torch.randn(M, K, device=dev).to(BF), (torch.randn(N, K, device=dev) / K ** 0.5).to(BF)
    wt = w.t().contiguous()
    res = torch.randn(M, N, device=dev).to(BF); aux = torch.randn(M, N, device=dev).to(BF)
    shift = torch.randn(N, device=dev)
    out = torch.empty(M, N, dtype=BF, device=dev)
    def fwd(tile):
        return ops.linear(x, w, shift, out=out, act=k.ACT_RELU, tile=tile, split_k=1)
    def fwdres(tile):
        return ops.linear(x, w, shift, out=out, res=res, act=k.ACT_RELU, tile=tile, split_k=1)
    def dgr(tile):
        k.gemm(M, N, K, k.A_ROWK, k.operand(x, K), k.B_KROW, k.operand(wt, N), out, N, res=res, ldr=N, act=k.ACT_MASK_POS, aux=aux, ldaux=N, tile=tile)
        return out
    for name, f in (("fwd", fwd), ("fwd+res", fwdres), ("dgrad", dgr)):
        ref = f(65).float().clone()
        row = []
        for tile in (65, 130, 136):
            try:
                dev_max = 0.0
                if tile == 136:
                    for rep in range(3):
                        out.zero_()
                        got = f(tile).float()
                        dev_max = max(dev_max, float(((got - ref).abs() / (ref.abs() + 1.0)).max()))
                t = timeit(lambda: f(tile), 20) * 1000
                row.append(f"{tile} {t:6.1f}" + (f" (dev {dev_max:.1e})" if tile == 136 else ""))
            except Exception as e:
                row.append(f"{tile}   n/a ({str(e)[-50:]})")
        fl = 2 * M * N * K
        print(f"{M:6d} {N:5d} {K:5d} {name:8s} " + "  ".join(row), flush=True)
