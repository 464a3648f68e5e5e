"""Few-channel 3x3 convolutions of the mask head at 160x160 x 800 maps: direct kernel (csrc/smallconv.hip) vs the
tiled implicit GEMM, against the HBM roofline (read input once + write output once).  GPU only."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from toist_amd import kernels as k, ops  # noqa: E402
from tools.bench_gemm import timeit  # noqa: E402

BF = torch.bfloat16
dev = torch.device("cuda")
Nb, H, W = 800, 160, 160
for C, Co in [(32, 16), (16, 8)]:
    x = torch.randn(Nb, H, W, C, device=dev).to(BF)
    w = (torch.randn(Co, 3, 3, C, device=dev) * 0.05).to(BF)
    b = torch.zeros(Co, device=dev)
    y = torch.empty(Nb, H, W, Co, dtype=BF, device=dev)
    dy = torch.randn(Nb, H, W, Co, device=dev).to(BF)
    dx = torch.empty(Nb, H, W, C, dtype=BF, device=dev)
    dw = torch.zeros(Co, 3, 3, C, device=dev)
    gb = Nb * H * W * (C + Co) * 2 / 1e9
    rows = [("fwd direct", lambda: ops.conv2d(x, w, pad=1, shift=b, out=y)), ("fwd tiled", lambda: ops.conv2d(x, w, pad=1, shift=b, out=y, tile=65)),
            ("dgrad direct", lambda: ops.conv2d_dgrad(dy, w, (H, W), pad=1, out=dx)),
            ("wgrad", lambda: ops.conv2d_wgrad(dy, x, (Co, 3, 3, C), pad=1, out=dw))]
    for name, fn in rows:
        ms = timeit(fn, 10)
        print(f"C{C}->Co{Co} {name:14s} {ms:7.3f} ms   {gb / ms * 1e3:7.0f} GB/s algorithmic ({100 * gb / ms * 1e3 / 8000:.0f} % of 8 TB/s)", flush=True)
