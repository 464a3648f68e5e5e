"""Tile / split-K sweep of the GEMM kernel on hot-path shapes (GPU only)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from toist_amd import kernels as k, ops  # noqa: E402
from tools.bench_gemm import timeit  # noqa: E402

BF = torch.bfloat16
dev = torch.device("cuda")
R = 256
TILES = [64, 65, 65 + 3 * R, 130, 129]


def row(name, fl, fn, splits=(0,)):
    out = []
    for t in TILES:
        for sp in splits:
            k.FORCE_TILE, k.FORCE_SPLIT = t, sp
            try:
                ms = timeit(fn, 20)
                out.append(f"{t & 255}r{t >> 8}{'/s%d' % sp if sp else ''}:{fl / ms / 1e9:5.0f}")
            except RuntimeError as e:
                out.append(f"{t}:ERR")
    k.FORCE_TILE = k.FORCE_SPLIT = 0
    print(f"{name:40s} " + " ".join(out), flush=True)


def lin(name, M, N, K):
    x, w = torch.randn(M, K, device=dev).to(BF), torch.randn(N, K, device=dev).to(BF)
    out = torch.empty(M, N, dtype=BF, device=dev)
    dy = torch.randn(M, N, device=dev).to(BF)
    dx = torch.empty(M, K, dtype=BF, device=dev)
    dw = torch.zeros(N, K, device=dev)
    fl = 2 * M * N * K
    row(name + " fwd", fl, lambda: ops.linear(x, w, out=out))
    wt = w.t()
    print(f"{name + ' fwd hipBLASLt (target only)':40s} {fl / timeit(lambda: torch.matmul(x, wt, out=out), 20) / 1e9:6.0f}", flush=True)
    row(name + " dgrad", fl, lambda: ops.linear_dgrad(dy, w, out=dx))
    for sp in (1, 4, 16):
        row(name + f" wgrad s{sp}", fl, lambda: ops.linear_wgrad(dy, x, out=dw), splits=(sp,))


def conv(name, Nb, H, W, C, Co, R, stride):
    pad = R // 2
    x = torch.randn(Nb, H, W, C, device=dev).to(BF)
    w = (torch.randn(Co, R, R, C, device=dev) * 0.05).to(BF)
    shift = torch.zeros(Co, device=dev)
    OH, OW = ops.conv_out_hw(H, W, R, R, stride, pad)
    y = torch.empty(Nb, OH, OW, Co, dtype=BF, device=dev)
    fl = 2 * Nb * OH * OW * Co * R * R * C
    dy = torch.randn(Nb, OH, OW, Co, device=dev).to(BF)
    dx = torch.empty(Nb, H, W, C, dtype=BF, device=dev)
    dw = torch.zeros(Co, R, R, C, device=dev)
    row(name + " fwd", fl, lambda: ops.conv2d(x, w, stride=stride, pad=pad, shift=shift, act=k.ACT_RELU, out=y))
    row(name + " dgrad", fl, lambda: ops.conv2d_dgrad(dy, w, (H, W), stride=stride, pad=pad, out=dx, act=k.ACT_MASK_POS, aux=x))
    for sp in (1, 4, 16):
        row(name + f" wgrad s{sp}", fl, lambda: ops.conv2d_wgrad(dy, x, (Co, R, R, C), stride=stride, pad=pad, out=dw), splits=(sp,))


print("columns: tile code (64=64x64x32 65=64x64x64 128=128x128x32 129=128x128x64 130=128x64x64) : TFLOP/s")
lin("square 4096", 4096, 4096, 4096)
lin("l1.conv3 M204800 N256 K64", 204800, 256, 64)
lin("l2.conv3 M51200 N512 K128", 51200, 512, 128)
lin("l3.conv1 M12800 N256 K1024", 12800, 256, 1024)
lin("l3.conv3 M12800 N1024 K256", 12800, 1024, 256)
lin("l4.conv1 M3200 N512 K2048", 3200, 512, 2048)
lin("enc ffn1 M3328 N2048 K256", 3328, 2048, 256)
lin("enc qk M3328 N512 K256", 3328, 512, 256)
lin("roberta M128 N768 K768", 128, 768, 768)
conv("l2 3x3 80x80 C128", 8, 80, 80, 128, 128, 3, 1)
conv("l3 3x3 40x40 C256", 8, 40, 40, 256, 256, 3, 1)
conv("l4 3x3 20x20 C512", 8, 20, 20, 512, 512, 3, 1)
