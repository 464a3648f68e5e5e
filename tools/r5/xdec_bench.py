"""Decoder forward (6 layers, B = 8, 100 queries, 416 memory tokens) as the per-op launches of toist_amd.tlayer vs the ONE XCD-resident launch of
csrc/xdec.hip: time per forward under hipGraph replay, and the in-kernel phase stamps of the fused launch (toist_xdec_desc.prof).
Usage (GPU box): python tools/r5/xdec_bench.py [--train]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import toist_amd  # noqa: E402
from toist_amd import harness, tlayer  # noqa: E402
from toist_amd import kernels as k  # noqa: E402
from tools.bench_gemm import timeit  # noqa: E402

BF = torch.bfloat16


def backward_profile(tr, memory, pos, key_pad, qe, B, S, dev):
    """forward + backward of the decoder program: per-op backward vs the one-launch data-gradient chain, and the backward launch's phase stamps"""
    L = 6
    mem = memory.clone().requires_grad_(True)
    g = torch.randn(L, B * 100, 256, device=dev).to(BF)

    def fb():
        mem.grad = None
        out = tr.decode_tokens(mem, pos, key_pad, qe, B, S)
        out.backward(g.view_as(out))

    res = {}
    for flag in (False, True):
        tlayer.XDEC, tlayer.XDEC_BWD = True, flag
        fb()
        torch.cuda.synchronize()
        res[flag] = timeit(fb, 10) * 1000.0
    k.xdec_check()
    print(f"decoder forward + backward (weight gradients included): per-op backward {res[False]:.1f} us, one-launch data-gradient chain {res[True]:.1f} us")
    k.XDEC_PROF = torch.zeros(256, L, 16, dtype=torch.int64, device=dev)
    tlayer.XDEC, tlayer.XDEC_BWD = True, True
    fb()
    torch.cuda.synchronize()
    st = k.XDEC_PROF.cpu().double() * 0.01
    k.XDEC_PROF = None
    t0 = st[:, L - 1, 0].min()
    print(f"backward launch: {float(st[:, 0, 13].max() - t0):.1f} us between the first and the last stamp")
    names = ["R0 norm4 bwd", "wait", "H dh + partial", "wait", "C fold + norm3 bwd + x W_oc", "wait", "D cross-attn bwd", "wait", "E fold dq + x W_q + norm1 bwd + x W_os", "wait",
             "F self-attn bwd", "wait", "G x W_in"]
    for layer in (L - 1, 2, 0):
        seg = st[:, layer, 1:14] - st[:, layer, :13]
        print(f"layer {layer}: start +{float(st[:, layer, 0].min() - t0):.1f} us | " + " | ".join(f"{n} {float(seg[:, i].mean()):.2f} (max {float(seg[:, i].max()):.2f})" for i, n in enumerate(names)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--train", action="store_true")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--tokens", type=int, default=416)
    ap.add_argument("--bwd", action="store_true", help="phase stamps of the backward launch (toist_xdec_bwd) and fwd + bwd timings")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    args = harness.default_args(device="cuda", contrastive_align_loss=True)
    model, _, _, _ = toist_amd.build_model(args)
    model.to(dev).train(a.train)
    tr = model.transformer
    B, S, Q, d = a.batch, a.tokens, 100, 256
    g = torch.Generator().manual_seed(1)
    memory = torch.randn(B * S, d, generator=g).to(dev).to(BF)
    pos = torch.randn(B * S, d, generator=g).to(dev).to(BF)
    key_pad = torch.zeros(B, S, dtype=torch.uint8, device=dev)
    key_pad[:, S - 5:] = 1
    qe = model.query_embed.weight
    k.SEED_DEV = torch.zeros(1, dtype=torch.int64, device=dev)

    def fwd():
        with torch.no_grad():
            return tr.decode_tokens(memory, pos, key_pad, qe, B, S)

    res = {}
    for flag in (False, True):
        tlayer.XDEC = flag
        tr._step = 0
        out = fwd().float()
        torch.cuda.synchronize()
        res[flag] = (timeit(fwd, 10) * 1000.0, out)
    k.xdec_check()
    rel = float((res[True][1] - res[False][1]).norm() / res[False][1].norm())
    print(f"decoder forward, B={B} Q={Q} S={S} train={a.train}: per-op launches {res[False][0]:.1f} us, XCD-resident launch {res[True][0]:.1f} us, outputs differ by {rel:.2e} (relative Frobenius)")
    # phase stamps of one eager fused forward
    L = 6
    k.XDEC_PROF = torch.zeros(256, L, 16, dtype=torch.int64, device=dev)
    tlayer.XDEC = True
    fwd()
    torch.cuda.synchronize()
    st = k.XDEC_PROF.cpu().double() * 0.01      # us
    k.XDEC_PROF = None
    live = st[:, 0, 0] > 0
    st = st[live]
    t0 = st[:, 0, 0].min()
    names = ["P1 q|k|v", "wait", "A rows (attn, norm1, cross, norm3)", "FFN weights + wait", "P6 hidden + linear2 partials", "wait", "P7 fold + norm4"]
    print(f"{int(live.sum())} workgroups stamped; whole launch {float(st[:, L - 1, 7].max() - t0):.1f} us between the first P1 and the last P7 stamp")
    own = st[:, 0, 8] > 0
    ao = st[own]
    for layer in (0, 3):
        ch = [("q rows + self-attention", 2, 8), ("W_os stream + sync", 8, 9), ("out_proj + norm1", 9, 10), ("query projection", 10, 11), ("cross-attention", 11, 12),
              ("W_oc stream + sync", 12, 13), ("out_proj + norm3", 13, 3)]
        print(f"layer {layer} phase A of the {int(own.sum())} row owners: " + " | ".join(f"{n} {float((ao[:, layer, j] - ao[:, layer, i]).mean()):.2f}" for n, i, j in ch))
    w0 = st[:, :, 14] > 0
    if bool(w0.any()):
        print("P6 of wave 0 (workgroup mean over layers): y3 fragments arrive after %.2f us, linear1 + ReLU + dropout + h stores %.2f us, linear2 partials + stores %.2f us" % (
            float((st[:, :, 14] - st[:, :, 4])[w0].mean()), float((st[:, :, 15] - st[:, :, 14])[w0].mean()), float((st[:, :, 5] - st[:, :, 15])[w0].mean())))
    if a.bwd:
        return backward_profile(tr, memory, pos, key_pad, qe, B, S, dev)
    for layer in range(L):
        seg = st[:, layer, 1:8] - st[:, layer, :7]
        row = " | ".join(f"{n} {float(seg[:, i].mean()):.2f} (max {float(seg[:, i].max()):.2f})" for i, n in enumerate(names))
        print(f"layer {layer}: start +{float(st[:, layer, 0].min() - t0):.1f} us | {row}")


if __name__ == "__main__":
    main()
