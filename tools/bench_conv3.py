"""3x3 stride-1 convolution: shared-halo kernel (tile code 131) vs the generic implicit-GEMM tiles (GPU only)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from toist_amd import kernels as k, ops  # noqa: E402
from tools.bench_gemm import timeit  # noqa: E402

BF = torch.bfloat16
dev = torch.device("cuda")
print("TFLOP/s by tile code (65 = 64x64x64 generic, 130 = 128x64x64 generic, 131 = shared-halo 128x64)")
for name, Nb, H, W, C, Co in [("l3 40x40 C256", 8, 40, 40, 256, 256), ("l4 20x20 C512", 8, 20, 20, 512, 512), ("40x40 C256 B=16", 16, 40, 40, 256, 256)]:
    x = torch.randn(Nb, H, W, C, device=dev).to(BF)
    w = (torch.randn(Co, 3, 3, C, device=dev) * 0.05).to(BF)
    shift = torch.zeros(Co, device=dev)
    y = torch.empty(Nb, H, W, Co, dtype=BF, device=dev)
    dy = torch.randn(Nb, H, W, Co, device=dev).to(BF)
    dx = torch.empty(Nb, H, W, C, dtype=BF, device=dev)
    fl = 2 * Nb * H * W * Co * 9 * C
    for label, fn in (("fwd", lambda: ops.conv2d(x, w, stride=1, pad=1, shift=shift, act=k.ACT_RELU, out=y)),
                      ("dgrad", lambda: ops.conv2d_dgrad(dy, w, (H, W), stride=1, pad=1, out=dx, act=k.ACT_MASK_POS, aux=x))):
        out = []
        for t in (65, 130, 131):
            k.FORCE_TILE = t
            ms = timeit(fn, 30)
            out.append(f"{t}:{fl / ms / 1e9:6.0f} ({1000 * ms:6.1f} us)")
        k.FORCE_TILE = 0
        print(f"{name:18s} {label:6s} " + "  ".join(out), flush=True)
