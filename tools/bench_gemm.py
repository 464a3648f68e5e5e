"""Micro-benchmark of the GEMM / implicit-conv kernel on the shapes of the TOIST hot path (GPU only).
Usage: python tools/bench_gemm.py [--iters 20]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from toist_amd import kernels as k, ops  # noqa: E402

BF = torch.bfloat16


def timeit(fn, iters):
    """GPU time per call: the launches are captured into a hipGraph and replayed, so the Python /
    ctypes launch cost (~10-20 us per call, larger than most hot-path kernels) is not measured."""
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        fn()
        with torch.cuda.graph(graph, stream=side):
            for _ in range(iters):
                fn()
    torch.cuda.current_stream().wait_stream(side)
    graph.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    graph.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    dev = torch.device("cuda")
    rows = []

    def lin(name, M, N, K, tile=0):
        x, w = torch.randn(M, K, device=dev).to(BF), torch.randn(N, K, device=dev).to(BF)
        out = torch.empty(M, N, dtype=BF, device=dev)
        ms = timeit(lambda: ops.linear(x, w, out=out, tile=tile), a.iters)
        rows.append((name, 2 * M * N * K, ms))
        dy = torch.randn(M, N, device=dev).to(BF)
        dx = torch.empty(M, K, dtype=BF, device=dev)
        ms = timeit(lambda: ops.linear_dgrad(dy, w, out=dx), a.iters)
        rows.append((name + " dgrad", 2 * M * N * K, ms))
        dw = torch.zeros(N, K, device=dev)
        ms = timeit(lambda: ops.linear_wgrad(dy, x, out=dw), a.iters)
        rows.append((name + " wgrad", 2 * M * N * K, ms))

    def conv(name, Nb, H, W, C, Co, R, stride):
        pad = R // 2
        x = torch.randn(Nb, H, W, C, device=dev).to(BF)
        w = (torch.randn(Co, R, R, C, device=dev) * 0.05).to(BF)
        shift = torch.zeros(Co, device=dev)
        OH, OW = ops.conv_out_hw(H, W, R, R, stride, pad)
        y = torch.empty(Nb, OH, OW, Co, dtype=BF, device=dev)
        fl = 2 * Nb * OH * OW * Co * R * R * C
        ms = timeit(lambda: ops.conv2d(x, w, stride=stride, pad=pad, shift=shift, act=k.ACT_RELU, out=y), a.iters)
        rows.append((name + " fwd", fl, ms))
        dy = torch.randn(Nb, OH, OW, Co, device=dev).to(BF)
        dx = torch.empty(Nb, H, W, C, dtype=BF, device=dev)
        ms = timeit(lambda: ops.conv2d_dgrad(dy, w, (H, W), stride=stride, pad=pad, out=dx, act=k.ACT_MASK_POS, aux=x), a.iters)
        rows.append((name + " dgrad", fl, ms))
        dw = torch.zeros(Co, R, R, C, device=dev)
        ms = timeit(lambda: ops.conv2d_wgrad(dy, x, (Co, R, R, C), stride=stride, pad=pad, out=dw), a.iters)
        rows.append((name + " wgrad", fl, ms))

    lin("square 4096", 4096, 4096, 4096)
    lin("l1.conv3 1x1 M204800 N256 K64", 204800, 256, 64)
    lin("l2.conv1 1x1 M51200 N128 K512", 51200, 128, 512)
    lin("l3.conv1 1x1 M12800 N256 K1024", 12800, 256, 1024)
    lin("l3.conv3 1x1 M12800 N1024 K256", 12800, 1024, 256)
    lin("l4.conv3 1x1 M3200 N2048 K512", 3200, 2048, 512)
    lin("enc ffn1 M3328 N2048 K256", 3328, 2048, 256)
    lin("enc ffn2 M3328 N256 K2048", 3328, 256, 2048)
    lin("enc qk M3328 N512 K256", 3328, 512, 256)
    conv("l1 3x3 160x160 C64", 8, 160, 160, 64, 64, 3, 1)
    conv("l2 3x3 80x80 C128", 8, 80, 80, 128, 128, 3, 1)
    conv("l3 3x3 40x40 C256", 8, 40, 40, 256, 256, 3, 1)
    conv("l4 3x3 20x20 C512", 8, 20, 20, 512, 512, 3, 1)
    conv("l3.0 3x3 s2 80->40 C256", 8, 80, 80, 256, 256, 3, 2)
    conv("stem 7x7 s2 C8", 8, 640, 640, 8, 64, 7, 2)
    print(f"{'shape':44s} {'ms':>9s} {'TFLOP/s':>9s}")
    for name, fl, ms in rows:
        print(f"{name:44s} {ms:9.4f} {fl / ms / 1e9:9.1f}")


if __name__ == "__main__":
    main()
