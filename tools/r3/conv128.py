"""Round 3: gemm128_kernel (tile 136) as a stride-1 convolution gather (forward) and transposed gather (data gradient) against the
dispatcher's choice (tile 0) on the 3x3 shapes of the step: us per launch (hipGraph replay) and max deviation from the 64 x 64 tile."""
import sys, os
sys.path.insert(0, os.getcwd())
import torch
from toist_amd import kernels as k, ops
from tools.bench_gemm import timeit
BF = torch.bfloat16
dev = torch.device("cuda")
#         N   H   W    C   Co dil
shapes = ((8, 80, 80, 128, 128, 1), (8, 40, 40, 256, 256, 1), (8, 20, 20, 512, 512, 1), (8, 40, 40, 256, 256, 2), (2, 37, 43, 128, 256, 1), (1, 25, 38, 64, 128, 1))
for Nb, H, W, C, Co, dil in shapes:
    x = torch.randn(Nb, H, W, C, device=dev).to(BF)
    w = (torch.randn(Co, 3, 3, C, device=dev) / (9 * C) ** 0.5).to(BF)
    dy = torch.randn(Nb, H, W, Co, device=dev).to(BF)
    aux = torch.randn(Nb, H, W, C, device=dev).to(BF)
    scale, shift = torch.rand(Co, device=dev) + 0.5, torch.randn(Co, device=dev)
    out = torch.empty(Nb, H, W, Co, dtype=BF, device=dev)
    dx = torch.empty(Nb, H, W, C, dtype=BF, device=dev)
    def fwd(tile):
        return ops.conv2d(x, w, pad=dil, dil=dil, scale=scale, shift=shift, act=k.ACT_RELU, out=out, tile=tile)
    def dgr(tile):
        return ops.conv2d_dgrad(dy, w, (H, W), pad=dil, dil=dil, act=k.ACT_MASK_POS, aux=aux, out=dx, tile=tile)
    for name, f in (("fwd", fwd), ("dgrad", dgr)):
        ref = f(65).float().clone()
        row = []
        for tile in (0, 65, 136):
            try:
                dm = 0.0
                if tile == 136:
                    for rep in range(3):
                        (out if name == "fwd" else dx).zero_()
                        got = f(tile).float()
                        dm = max(dm, float(((got - ref).abs() / (ref.abs() + 1.0)).max()))
                t = timeit(lambda: f(tile), 20) * 1000
                row.append(f"{tile} {t:6.1f}" + (f" (dev {dm:.1e})" if tile == 136 else ""))
            except Exception as e:
                row.append(f"{tile}   n/a ({str(e)[-60:]})")
        fl = 2 * Nb * H * W * Co * 9 * C
        print(f"{Nb}x{H}x{W} C{C}->{Co} d{dil} {name:6s} " + "  ".join(row), flush=True)
