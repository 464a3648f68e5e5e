"""CU-masked streams (hipExtStreamCreateWithCUMask): does the mask hold for plain launches and for a replayed single-chain hipGraph, and
how is a contiguous run of mask bits spread over the chip?  A chip-filling GEMM on a stream with n of 256 CUs should take ~256 / n times
as long.  GPU only."""
import sys

import torch

import ctypes

_HIP = ctypes.CDLL("libamdhip64.so")


def masked_stream(dev, n_cus, first=0):
    """hipExtStreamCreateWithCUMask, bits first .. first + n_cus - 1, wrapped for torch"""
    words = (ctypes.c_uint32 * 8)(*([0] * 8))
    for b in range(first, first + n_cus):
        words[(b % 256) >> 5] |= 1 << (b & 31)
    handle = ctypes.c_void_p()
    rc = _HIP.hipExtStreamCreateWithCUMask(ctypes.byref(handle), 8, words)
    assert rc == 0 and handle.value, rc
    return torch.cuda.ExternalStream(handle.value, device=dev)


dev = torch.device("cuda")
a = torch.randn(8192, 4096, device=dev, dtype=torch.bfloat16)
b = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16) * 0.02
out = torch.empty(8192, 4096, device=dev, dtype=torch.bfloat16)


def timed(stream, fn, reps=10):
    with torch.cuda.stream(stream):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(reps):
            fn()
        e1.record(stream)
        torch.cuda.synchronize()
    return 1000.0 * e0.elapsed_time(e1) / reps


mm = lambda: torch.mm(a, b, out=out)
base = timed(torch.cuda.current_stream(), mm)
print(f"mm 8192 x 4096 x 4096 on the default stream: {base:8.1f} us")
cap = torch.cuda.Stream()
g = torch.cuda.CUDAGraph()
with torch.cuda.stream(cap):
    mm()
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=cap):
        mm()
        mm()
torch.cuda.current_stream().wait_stream(cap)
print(f"graph of two, replayed on the default stream: {timed(torch.cuda.current_stream(), g.replay) / 2:8.1f} us per mm")
for n, first in ((128, 0), (64, 0), (32, 0), (32, 32), (32, 100), (16, 0), (8, 0)):
    s = masked_stream(dev, n, first)
    t = timed(s, mm)
    tg = timed(s, g.replay) / 2
    print(f"  {n:3d} CUs (bits {first:3d} ..): plain {t:8.1f} us = {t / base:5.2f} x  (256 / n = {256 / n:5.2f}) | graph replayed on it {tg:8.1f} us per mm")
