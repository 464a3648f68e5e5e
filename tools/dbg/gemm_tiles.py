"""Tile sweep over the hot GEMM shapes of the batch-8 640^2 step: forward (B_ROWK), data gradient (B_KROW), 3x3 convs."""
import sys, os
sys.path.insert(0, os.getcwd())
import torch
from toist_amd import kernels as k, ops
from tools.bench_gemm import timeit
BF = torch.bfloat16
dev = torch.device("cuda")
TILES = (65, 130, 134, 129)

def sweep(label, fn):
    row = []
    ref = None
    for t in TILES:
        k.FORCE_TILE = t
        try:
            out = fn().float()
            if ref is None: ref = out
            err = float((out - ref).abs().max())
            us = timeit(fn, 20) * 1000
            row.append(f"{t}:{us:6.1f}" + ("" if err == 0 else f"(!{err:.1e})"))
        except Exception as e:
            row.append(f"{t}: ERR")
    k.FORCE_TILE = 0
    print(f"{label:44s}", " ".join(row), flush=True)

def lin(M, N, K):
    x, w = torch.randn(M, K, device=dev).to(BF), (torch.randn(N, K, device=dev) * 0.05).to(BF)
    wt = w.t().contiguous()
    out = torch.empty(M, N, dtype=BF, device=dev)
    res = torch.randn(M, N, device=dev).to(BF); shift = torch.zeros(N, device=dev)
    sweep(f"fwd   ({M},{N},{K})", lambda: ops.linear(x, w, shift, out=out, res=res, act=k.ACT_RELU))
    def dg():
        k.gemm(M, N, K, k.A_ROWK, k.operand(x, K), k.B_KROW, k.operand(wt, N), out, N, res=res, ldr=N)
        return out
    sweep(f"dgrad ({M},{N},{K})", dg)

def conv(Nb, H, C, Co, stride=1):
    x = torch.randn(Nb, H, H, C, device=dev).to(BF)
    w = (torch.randn(Co, 3, 3, C, device=dev) * 0.02).to(BF)
    shift = torch.zeros(Co, device=dev); scale = torch.ones(Co, device=dev)
    sweep(f"conv3 fwd  {Nb}x{H}x{H} {C}->{Co} s{stride}", lambda: ops.conv2d(x, w, stride=stride, pad=1, scale=scale, shift=shift, act=k.ACT_RELU))
    OH = (H + 2 - 3) // stride + 1
    dy = torch.randn(Nb, OH, OH, Co, device=dev).to(BF)
    sweep(f"conv3 dgrad {Nb}x{H}x{H} {C}->{Co} s{stride}", lambda: ops.conv2d_dgrad(dy, w, (H, H), stride=stride, pad=1))

TILES = (0, 65, 131, 134, 130)
conv(8, 40, 256, 256)
conv(8, 80, 128, 128)
conv(8, 20, 512, 512)
conv(8, 160, 64, 64)
