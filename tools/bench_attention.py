"""MFMA utilisation of the encoder-decoder attention at batch 8 (BASELINE.json north_star target), measured on the product code
path of round 2: the fused attention blocks of toist_amd.engine (packed in_proj as one launch on two inputs, flash-style core with
row statistics only, out-proj + dropout + residual, grouped decoder K / V projections, merged data gradients; forward + backward,
dropout 0.1) captured in a hipGraph and replayed.  GPU only.

FLOP accounting: cores 4*Sq*Sk*d, projections 2*(2*Sq + 2*Sk)*d*d per image, x3 for forward + backward.
Next to every case: the number of kernel launches of the replayed graph, the time the same number of EMPTY launches takes
(the launch-latency floor of this launch count) and the utilisation that floor alone would allow.

Round 4: the default path is the product path of toist_amd.tlayer -- attn2 cores (csrc/attn2.hip), out_proj + dropout + residual +
LayerNorm as one row-complete launch, and in the backward pass [in_proj data gradient + fold of the key-split dQ shares + residual
gradient + LayerNorm backward] as one launch (csrc/tlayer.hip).  A block here therefore ALSO contains one LayerNorm forward and one
LayerNorm backward that the round-3 block (`--v1`: toist_amd.engine blocks) left to separate launches."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from toist_amd import engine, kernels as k  # noqa: E402

dev = torch.device("cuda")
BF = torch.bfloat16
V1 = "--v1" in sys.argv
B, d, H, L = 8, 256, 8, 6
PEAK = 2500.0


def params(n_layers):
    """[(Win, bin, Wo, bo)] ParamViews with gradient slots, one set per layer."""
    torch.manual_seed(0)
    out = []
    for _ in range(n_layers):
        mk = lambda *s: (torch.randn(*s, device=dev) * 0.05)
        ts = [mk(3 * d, d), mk(3 * d), mk(d, d), mk(d)]
        out.append(tuple(engine.ParamView(t.to(BF) if t.dim() == 2 else None, torch.zeros_like(t), t) for t in ts))
    return out


def graphed(step, reps=50):
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
    return 1000.0 * e0.elapsed_time(e1) / reps


class Counter:
    """counts toist kernel launches of one step() by wrapping the ctypes entry points used here"""

    def __init__(self):
        self.n = 0

    def __enter__(self):
        from toist_amd import _lib
        self.lib = _lib.lib()
        self.saved = {}
        for name in ("toist_gemm_bf16", "toist_add_bf16", "toist_splitk_reduce_batch", "toist_group_fill", "toist_layernorm_fwd",
                     "toist_attn2_fwd", "toist_attn2_bwd", "toist_rowgemm", "toist_layernorm_bwd"):
            fn = getattr(self.lib, name)
            self.saved[name] = fn

            def wrap(*a, _fn=fn, _name=name):
                self.n += 1
                return _fn(*a)

            setattr(self.lib, name, wrap)
        return self

    def __exit__(self, *exc):
        for name, fn in self.saved.items():
            setattr(self.lib, name, fn)


def empty_launch_us(n):
    x = torch.zeros(1024, device=dev).to(BF)
    y = torch.empty_like(x)

    def step():
        for _ in range(n):
            k.add(x, x, y)

    return graphed(step)


def run(name, Sq, Sk, kind):
    key_pad = torch.zeros(B, Sk, dtype=torch.uint8, device=dev)
    k.SEED_DEV = torch.zeros(1, dtype=torch.int64, device=dev)
    P = params(L)
    x = torch.randn(B * Sq, d, device=dev).to(BF)
    e = torch.randn(B * Sq, d, device=dev).to(BF)
    mem = torch.randn(B * Sk, d, device=dev).to(BF)
    pos = torch.randn(B * Sk, d, device=dev).to(BF)

    def step():
        tape = engine.Tape(training=True, drop_p=float(os.environ.get("TOIST_BENCH_ATTN_DROP", "0.1")), seed=1, group_wgrads=True)
        outs = []
        if kind == "cross":
            m = engine.Var(mem)
            mem_e = torch.empty_like(mem)
            k.add(mem, pos, mem_e)
            kv, dkv = engine.cross_kv_projections(tape, m, mem_e, [(p[0], p[1]) for p in P])
        for i, (Wi, bi, Wo, bo) in enumerate(P):          # six independent blocks (one per layer), as a step runs them
            xv = engine.Var(x)
            xe = torch.empty_like(x)
            k.add(x, e, xe)                                 # stands for the LayerNorm that emits x + pos in the model
            if kind == "cross":
                o = engine.cross_attention_block(tape, xv, xe, Wi.rows(0, d), bi.rows(0, d), kv, dkv, i * 2 * d, Wo, bo, key_pad, B, Sq, Sk, H)
            else:
                o = engine.self_attention_block(tape, xv, xe, Wi, bi, Wo, bo, key_pad if kind == "enc" else None, B, Sq, H)
            o.grad = torch.ones_like(o.data)
            outs.append(o)
        tape.backward()

    def step2():
        """the same blocks on toist_amd.tlayer's kernels: [q|k|v GEMM] [attn2 forward] [out_proj + dropout + residual + LayerNorm] forward;
        [out_proj dgrad] [attn2 backward] [in_proj dgrad + dQ fold + residual gradient + LayerNorm backward] + grouped weight gradients"""
        from toist_amd import ops, tlayer
        tape = engine.Tape(training=True, drop_p=float(os.environ.get("TOIST_BENCH_ATTN_DROP", "0.1")), seed=1, group_wgrads=True)
        p = tape.drop_p
        M = B * Sq
        if kind == "cross":
            m = engine.Var(mem)
            mem_e = torch.empty_like(mem)
            k.add(mem, pos, mem_e)
            kv, dkv = engine.cross_kv_projections(tape, m, mem_e, [(q_[0], q_[1]) for q_ in P])
        saved = []
        for i, (Wi, bi, Wo, bo) in enumerate(P):
            xe = torch.empty_like(x)
            k.add(x, e, xe)
            if kind == "cross":
                qb = ops.linear(xe, Wi.rows(0, d).w, bi.rows(0, d).f32)
                kb, vb = kv[:, i * 2 * d:i * 2 * d + d], kv[:, i * 2 * d + d:(i + 1) * 2 * d]
            else:
                qkv = tlayer._qkv(xe, x, Wi, bi, M, d)
                qb, kb, vb = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]
            ctx = torch.empty(M, d, dtype=BF, device=dev)
            core = tlayer._core(tape, qb, kb, vb, key_pad if kind != "dec" else None, B, Sq, Sk, H, ctx, p, tape.next_seed())
            ln = tlayer._ln_fwd(ctx, Wo, bo, x, LN[i][0], LN[i][1], tape)
            saved.append((xe, ctx, core, ln, Wi, bi, Wo, bo))
        prev = None
        for i in reversed(range(L)):
            xe, ctx, core, ln, Wi, bi, Wo, bo = saved[i]
            if prev is None:
                tlayer._ln_bwd_alone(ln, ones, p)           # the gradient of the last block's output arrives from outside
            dctx = tlayer._outproj_bwd(tape, ln, ctx, Wo, bo, p)
            if kind == "cross":
                dq = torch.empty(M, d, dtype=BF, device=dev)
                part = core(dctx, dq, dkv[:, i * 2 * d:i * 2 * d + d], dkv[:, i * 2 * d + d:(i + 1) * 2 * d])
                a, w, Kk = dq, Wi.rows(0, d).w, d
                tape.linear_wgrad(dq, xe, Wi.rows(0, d), bi.rows(0, d))
            else:
                dqkv = torch.empty(M, 3 * d, dtype=BF, device=dev)
                part = core(dctx, dqkv[:, :d], dqkv[:, d:2 * d], dqkv[:, 2 * d:])
                a, w, Kk = dqkv, Wi.w, 3 * d
                tape.linear_wgrad(dqkv[:, :2 * d], xe, Wi.rows(0, 2 * d), bi.rows(0, 2 * d))
                tape.linear_wgrad(dqkv[:, 2 * d:], x, Wi.rows(2 * d, 3 * d), bi.rows(2 * d, 3 * d))
            below = saved[i - 1][3] if i > 0 else saved[L - 1][3]      # a LayerNorm to run backward through (the block below's; any at the bottom)
            tlayer._ln_bwd_fused(below, a, w, p, res=ln.dz, fold=part, fold_cols=d, K=Kk)
            prev = ln
        tape.backward()

    if not V1:
        # one LayerNorm per block, as in the model (shared parameters would force a fold of the d gamma / d beta partials per block)
        LN = [(engine.ParamView(None, torch.zeros(d, device=dev), torch.ones(d, device=dev)), engine.ParamView(None, torch.zeros(d, device=dev), torch.zeros(d, device=dev)))
              for _ in range(L)]
        ones = torch.ones(B * Sq, d, device=dev).to(BF)
        step = step2
    with Counter() as c:
        step()
    launches = c.n
    us = graphed(step) / L
    flop = 3.0 * B * (4.0 * Sq * Sk * d + 2.0 * (2 * Sq + 2 * Sk) * d * d)
    floor = empty_launch_us(launches) / L
    return {"case": name, "Sq": Sq, "Sk": Sk, "us_fwd_bwd": round(us, 1), "gflop": round(flop / 1e9, 2), "tflops": round(flop / us / 1e6, 1),
            "mfma_frac": round(flop / us / 1e6 / PEAK, 4), "launches_per_block": round(launches / L, 1), "empty_launch_floor_us": round(floor, 1),
            "mfma_frac_if_only_launch_bound": round(flop / floor / 1e6 / PEAK, 4)}


rows = [run("encoder self-attention (S=416)", 416, 416, "enc"), run("decoder cross-attention image+text (Q=100, S=416)", 100, 416, "cross"),
        run("decoder self-attention (Q=100)", 100, 100, "dec")]
tot_us = 6 * sum(r["us_fwd_bwd"] for r in rows)
tot_fl = 6 * sum(r["gflop"] for r in rows)
print(json.dumps({"batch": B, "layers": "6+6", "dropout": 0.1, "launch": "hipGraph replay", "path": "engine.self_attention_block / cross_attention_block / cross_kv_projections (round-3 blocks)" if V1 else
                  "toist_amd.tlayer: attn2 cores + row-complete out_proj / in_proj-dgrad launches (each block also holds one LayerNorm forward + backward)",
                  "cases": rows, "all_attention_per_step": {"ms": round(tot_us / 1e3, 3), "tflops": round(tot_fl / tot_us * 1e3, 1),
                                                            "mfma_frac": round(tot_fl / tot_us * 1e3 / PEAK, 4)}}, indent=1))
