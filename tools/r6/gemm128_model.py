"""Round 6, VERDICT r5 item 3(a): what could an XCD-local stream-K buy gemm128_kernel (tile 136)?  The kernel runs one 128 x 128 tile per CU, so a launch of
t <= 256 tiles with T k-tiles should cost fixed + T * per_k_tile whatever t is.  Timed (hipGraph replay, us per launch) on the layer-3 / layer-4 3x3
shapes with 100 / 200 / 256 tiles and with half the reduction depth: the fit gives the fixed cost and the k-tile time, and the BEST case of any
scheme that deals the 200 x T (tile, k-tile) units evenly over 256 CUs is fixed + T * 200 / 256 * per_k_tile (before its own fold / flag costs)."""
import os
import sys
sys.path.insert(0, os.getcwd())
import torch
from toist_amd import kernels as k, ops
from tools.bench_gemm import timeit
BF = torch.bfloat16
dev = torch.device("cuda")
rows = []
#            N   H   W    C   Co   what
for Nb, H, W, C, Co, what in ((8, 40, 40, 256, 256, "layer 3 3x3: 200 tiles, 36 k-tiles"), (4, 40, 40, 256, 256, "100 tiles, 36 k-tiles"),
                              (8, 32, 64, 256, 256, "256 tiles, 36 k-tiles"), (8, 40, 40, 128, 256, "200 tiles, 18 k-tiles"),
                              (8, 32, 64, 128, 256, "256 tiles, 18 k-tiles"), (8, 20, 20, 512, 512, "layer 4 3x3: 100 tiles, 72 k-tiles"),
                              (8, 20, 20, 256, 512, "100 tiles, 36 k-tiles"), (8, 80, 80, 128, 128, "layer 2 3x3: 400 tiles, 18 k-tiles")):
    x = torch.randn(Nb, H, W, C, device=dev).to(BF)
    w = (torch.randn(Co, 3, 3, C, device=dev) / (9 * C) ** 0.5).to(BF)
    scale, shift = torch.rand(Co, device=dev) + 0.5, torch.randn(Co, device=dev)
    out = torch.empty(Nb, H, W, Co, dtype=BF, device=dev)
    f = lambda: ops.conv2d(x, w, pad=1, scale=scale, shift=shift, act=k.ACT_RELU, out=out, tile=136)
    f()
    us = timeit(f, 30) * 1000
    tiles = ((Nb * H * W + 127) // 128) * (Co // 128)
    T = 9 * C // 64
    fl = 2.0 * Nb * H * W * Co * 9 * C
    rows.append((what, tiles, T, us))
    print(f"{what:42s} tiles {tiles:4d}  k-tiles {T:3d}  {us:6.1f} us  {fl / us / 1e6:6.0f} TFLOP/s", flush=True)
d = {(t, T): us for _, t, T, us in rows}
if (200, 36) in d and (200, 18) in d:
    per = (d[(200, 36)] - d[(200, 18)]) / 18
    fixed = d[(200, 36)] - 36 * per
    print(f"fit at 200 tiles: fixed {fixed:.1f} us + {per:.3f} us per k-tile; best case of an even deal over 256 CUs: {fixed + 36 * 200 / 256 * per:.1f} us "
          f"(today {d[(200, 36)]:.1f}; VERDICT's kill line: 21 us)")
