"""Round 3: gemm128w_kernel (tile 137) on the weight gradients of the step -- single and grouped, 1x1 and 3x3 -- against the 64 x 64
tiles (reference values) and, with a KNOBS=1 build and TOIST_GEMM128W=0, against the tiles the host picked before: us per call
(hipGraph replay incl. the folds of split launches) and max |diff| / max |ref|."""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.getcwd()))
import torch
from toist_amd import kernels as k, ops
from tools.bench_gemm import timeit
BF = torch.bfloat16
dev = torch.device("cuda")
torch.manual_seed(0)


def mk(Nb, H, W, C, Co, R, n):
    items = []
    for _ in range(n):
        dy = torch.randn(Nb, H, W, Co, device=dev).to(BF)
        x = torch.randn(Nb, H, W, C, device=dev).to(BF)
        out = torch.randn(Co, R, R, C, device=dev) * 0.1
        rs = torch.rand(Co, device=dev) + 0.5
        items.append((dy, x, out, rs))
    return items


def run_single(items, R, dil, tile):
    k.FORCE_TILE = tile
    for dy, x, out, rs in items:
        ops.conv2d_wgrad(dy, x, out.shape, pad=dil * (R // 2), dil=dil, out=out, rscale=rs, defer=True, accumulate=True)
    k.flush_reductions()
    k.FORCE_TILE = 0


def run_group(items, R, dil, tile):
    old = ops.GROUP_TILE
    ops.GROUP_TILE = tile
    ops.conv2d_wgrad_group(items, items[0][2].shape, pad=dil * (R // 2), dil=dil, accumulate=True)
    k.flush_reductions()
    ops.GROUP_TILE = old


#        Nb   H   W    C    Co  R dil  n
cases = ((8, 40, 40, 1024, 256, 1, 1, 1), (8, 40, 40, 256, 1024, 1, 1, 1), (8, 40, 40, 256, 256, 3, 1, 1),
         (8, 40, 40, 1024, 256, 1, 1, 22), (8, 40, 40, 256, 1024, 1, 1, 22), (8, 40, 40, 256, 256, 3, 1, 22),
         (8, 80, 80, 512, 128, 1, 1, 3), (8, 80, 80, 128, 512, 1, 1, 3), (8, 80, 80, 128, 128, 3, 1, 3),
         (8, 20, 20, 2048, 512, 1, 1, 2), (8, 20, 20, 512, 2048, 1, 1, 2), (8, 20, 20, 512, 512, 3, 1, 2),
         (8, 40, 40, 256, 256, 3, 2, 2), (2, 37, 43, 128, 128, 3, 1, 2), (4, 24, 16, 128, 256, 3, 1, 1), (8, 52, 8, 256, 768, 1, 1, 1))
for Nb, H, W, C, Co, R, dil, n in cases:
    items = mk(Nb, H, W, C, Co, R, n)
    run = run_single if n == 1 else run_group
    base = [it[2].clone() for it in items]
    def reset():
        for it, b in zip(items, base):
            it[2].copy_(b)
    reset(); run(items, R, dil, 65)
    ref = [it[2].clone() for it in items]
    reset(); run(items, R, dil, 0)
    got = [it[2].clone() for it in items]
    err = max(float((g - r).abs().max() / r.abs().max()) for g, r in zip(got, ref))
    t_ref = timeit(lambda: run(items, R, dil, 65), 10) * 1000
    t_new = timeit(lambda: run(items, R, dil, 0), 10) * 1000
    gf = 2.0 * Nb * H * W * C * Co * R * R * n / 1e9
    print(f"{n:2d} x [{Nb}x{H}x{W}] C{C}->{Co} {R}x{R} d{dil}: 64x64 tiles {t_ref:7.1f} us | dispatcher {t_new:7.1f} us = {gf / t_new / 1e-3:6.0f} TFLOP/s | rel err {err:.1e}", flush=True)
