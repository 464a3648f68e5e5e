"""MFMA utilisation of the encoder-decoder attention at batch 8 (BASELINE.json north_star target), measured on the
product code path: engine.attention forward + backward (projections, QK^T, masked softmax with dropout, PV, output
projection, residual) captured in a hipGraph and replayed.  GPU only.

FLOP accounting: cores 4*Sq*Sk*d, projections 2*(2*Sq + 2*Sk)*d*d per image, x3 for forward+backward."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from toist_amd import engine, kernels as k  # noqa: E402

dev = torch.device("cuda")
BF = torch.bfloat16
B, d, H = 8, 256, 8
PEAK = 2500.0


def run(name, Sq, Sk, self_attn, reps=50):
    torch.manual_seed(0)
    mk = lambda *s: (torch.randn(*s, device=dev) * 0.05)
    Wi, bi, Wo, bo = mk(3 * d, d), mk(3 * d), mk(d, d), mk(d)
    views = {}
    flat = torch.zeros(Wi.numel() + bi.numel() + Wo.numel() + bo.numel(), device=dev)
    off = 0
    for n, t in (("Wi", Wi), ("bi", bi), ("Wo", Wo), ("bo", bo)):
        g = flat[off:off + t.numel()].view(t.shape)
        off += t.numel()
        views[n] = engine.ParamView(t.to(BF) if t.dim() == 2 else None, g, t)
    x = (torch.randn(B * Sq, d, device=dev)).to(BF)
    mem = x if self_attn else torch.randn(B * Sk, d, device=dev).to(BF)
    key_pad = torch.zeros(B, Sk, dtype=torch.uint8, device=dev)
    k.SEED_DEV = torch.zeros(1, dtype=torch.int64, device=dev)

    def step():
        tape = engine.Tape(training=True, drop_p=0.1, seed=1)
        q = engine.Var(x)
        kv = q if self_attn else engine.Var(mem)
        Wv, bv = views["Wi"], views["bi"]
        out = engine.attention(tape, q, kv, kv, (Wv.rows(0, d), bv.rows(0, d)), (Wv.rows(d, 2 * d), bv.rows(d, 2 * d)),
                               (Wv.rows(2 * d, 3 * d), bv.rows(2 * d, 3 * d)), views["Wo"], views["bo"], q, key_pad, B, Sq, Sk, H,
                               packed_qk=(Wv.rows(0, 2 * d), bv.rows(0, 2 * d)) if self_attn else None)
        out.grad = torch.ones_like(out.data)
        tape.backward()

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            step()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=s):
            step()
    torch.cuda.current_stream().wait_stream(s)
    graph.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        graph.replay()
    e1.record()
    torch.cuda.synchronize()
    us = 1000.0 * e0.elapsed_time(e1) / reps
    flop = 3.0 * B * (4.0 * Sq * Sk * d + 2.0 * (2 * Sq + 2 * Sk) * d * d)
    return {"case": name, "Sq": Sq, "Sk": Sk, "us_fwd_bwd": round(us, 1), "gflop": round(flop / 1e9, 2), "tflops": round(flop / us / 1e6, 1),
            "mfma_frac": round(flop / us / 1e6 / PEAK, 4)}


rows = [run("encoder self-attention (S=416)", 416, 416, True), run("decoder cross-attention image+text (Q=100, S=416)", 100, 416, False),
        run("decoder self-attention (Q=100)", 100, 100, True)]
tot_us = 6 * sum(r["us_fwd_bwd"] for r in rows)
tot_fl = 6 * sum(r["gflop"] for r in rows)
print(json.dumps({"batch": B, "layers": "6+6", "dropout": 0.1, "launch": "hipGraph replay", "cases": rows,
                  "all_attention_per_step": {"ms": round(tot_us / 1e3, 3), "tflops": round(tot_fl / tot_us * 1e3, 1),
                                             "mfma_frac": round(tot_fl / tot_us * 1e3 / PEAK, 4)}}, indent=1))
