"""Encoder / decoder layer programs on the row-complete sub-layer kernels (csrc/tlayer.hip).

Same arithmetic as /root/reference/models/transformer.py:290-304 (TransformerEncoderLayer.forward_post) and :362-408
(TransformerDecoderLayer.forward_post) with d_model = 256, laid out for the launch count:

  forward, encoder layer      packed q|k|v GEMM -> attention core -> [out_proj + dropout + residual + norm1] -> linear1 + ReLU + dropout
                              -> [linear2 + dropout + residual + norm2 (+ pos for the next layer's q, k)]                 5 launches (was 7)
  backward, encoder layer     linear2 dgrad (ReLU / dropout mask) -> [linear1 dgrad + residual gradient + norm1 backward + dropout mask]
                              -> out_proj dgrad -> attention core backward -> [in_proj dgrad + residual gradient + norm2 backward of
                              the layer BELOW]                                                                            5 launches (was 8)
  the decoder layer likewise (norm1 / norm3 / norm4, the shared final norm stays one launch over all six layer outputs).

A LayerNorm's backward is computed by the kernel that produces the gradient of its OUTPUT (the consumer's data-gradient GEMM), so the
gradient of a LayerNorm input appears directly; every stand-alone LayerNorm launch of the two stacks disappears except the ones whose
output gradient arrives from outside the program (last encoder layer, last decoder layer) and the shared final decoder norm.
Weight gradients go to the tape's grouped launches exactly as in toist_amd.engine."""
import math
import os
from types import SimpleNamespace

import torch

from . import engine, ops
from . import kernels as k
from .knobs import knob

BF16 = torch.bfloat16
ENABLED = knob("TOIST_ROWS", True)          # tests flip this to compare with the per-op path of toist_amd.engine


FUSE_FWD_MAX_K = knob("TOIST_ROWS_FWD_MAX_K", 768)      # forward sub-layers with a longer reduction run as GEMM + LayerNorm launch ...
FUSE_FWD_MAX_M = knob("TOIST_ROWS_FWD_MAX_M", 0)        # ... (a row-count exception for the decoder's 800 queries was measured and dropped, see _ln_fwd)
XDEC_BWD = knob("TOIST_XDEC_BWD", True)     # ... and the data-gradient chain of the decoder backward as one launch (toist_xdec_bwd)
# decoder forward as ONE XCD-resident launch (csrc/xdec.hip) when the shape and the device allow it.  TOIST_XDEC=0 is a PRODUCT switch (read without
# TOIST_KNOBS): the opt-out for a GPU this process does not own alone -- a second process, a CU-masked queue
XDEC = os.environ.get("TOIST_XDEC", "1") != "0"


def supported(d, H, Sk):
    return ENABLED and engine.FUSED_BLOCKS and d == 256 and d // H == 32


def _core(tape, qb, kb, vb, key_pad, B, Sq, Sk, H, ctx, p, seed_p):
    """attention core forward; returns core_bwd(dctx, dq, dk, dv) -> None, or the bf16 [splits, B*Sq, 256] partial sums of dq that the
    consumer of dq must fold (kernels.rowgemm(fold=...)) when the keys of a head are owned by several workgroups"""
    dh = qb.shape[1] // H
    scale = 1.0 / math.sqrt(dh)
    lse = torch.empty(B * H, Sq, 2, dtype=torch.float32, device=qb.device)
    k.attn2_fwd(qb, kb, vb, key_pad, B, H, Sq, Sk, dh, scale, p, seed_p, ctx, lse)
    return _core_bwd_of(qb, kb, vb, key_pad, B, Sq, Sk, H, ctx, p, seed_p, lse)


def _core_bwd_of(qb, kb, vb, key_pad, B, Sq, Sk, H, ctx, p, seed_p, lse):
    """backward closure of an attn2-convention forward that has already run (attn2_fwd above, or the XCD-resident decoder launch)"""
    dh = qb.shape[1] // H
    scale = 1.0 / math.sqrt(dh)
    splits = k.attn2_splits(Sk)

    def core_bwd(dctx, dq, dk, dv):
        part = torch.empty(splits, B * Sq, H * dh, dtype=BF16, device=qb.device) if splits > 1 else None
        k.attn2_bwd(qb, kb, vb, ctx, dctx, lse, key_pad, B, H, Sq, Sk, dh, scale, p, seed_p, dq if splits == 1 else None, dk, dv, dq_part=part)
        return part

    return core_bwd


def _ln_fwd(ctx_in, W, b, res, gamma, beta, tape, add=None, y=None, eps=1e-5):
    """y = LayerNorm(res + dropout(ctx_in W^T + b)) in one launch -> namespace with y, y2 (= y + add), z, mean, rstd, seed"""
    M = ctx_in.shape[0]
    dev = ctx_in.device
    p = tape.drop_p
    s = SimpleNamespace()
    s.seed = tape.next_seed() if p > 0 else 0
    s.z = torch.empty(M, 256, dtype=BF16, device=dev)
    s.y = torch.empty(M, 256, dtype=BF16, device=dev) if y is None else y
    s.y2 = torch.empty(M, 256, dtype=BF16, device=dev) if add is not None else None
    s.mean = torch.empty(M, dtype=torch.float32, device=dev)
    s.rstd = torch.empty(M, dtype=torch.float32, device=dev)
    s.gamma, s.beta = gamma, beta
    if ctx_in.shape[1] <= FUSE_FWD_MAX_K or M <= FUSE_FWD_MAX_M:
        k.rowgemm(ctx_in, W.w, s.y, b_kind=k.B_ROWK, epi=k.ROW_LN_FWD, bias=b.f32, res=res, drop_p=p, drop_seed=s.seed, gamma=gamma.f32, beta=beta.f32,
                  eps=eps, z=s.z, mean=s.mean, rstd=s.rstd, add=add, out2=s.y2)
    else:
        # deep reductions (linear2, K = 2048): a row-complete block streams the whole 1 MB weight through ONE CU (~18 us whatever the row
        # count, profiles/r04_rowgemm_us.txt); the tiled GEMM spreads that stream over the chip and GEMM + LayerNorm launch stay ahead:
        # 19.4 vs 22.6 us at 3328 rows, 14.8 vs 18.6 us at the decoder's 800 rows (in the step: -33 us of forward, profiles/r04_rowgemm_us.txt)
        kw = dict(drop_where=1, drop_p=p, drop_seed=s.seed) if p > 0 else {}
        ops.linear(ctx_in, W.w, b.f32, res=res, out=s.z, **kw)
        k.layernorm_fwd(s.z, gamma.f32, beta.f32, eps, s.y, s.mean, s.rstd, add=add, y2=s.y2)
    s.dz = s.dzd = None       # set by the consumer's fused backward launch
    return s


def _ln_bwd_alone(ln, g, p):
    """stand-alone LayerNorm backward (the gradient of the output arrived from outside the fused chain)"""
    ln.dz = torch.empty_like(g)
    ln.dzd = torch.empty_like(g) if p > 0 else None
    k.layernorm_bwd(g, ln.z, ln.mean, ln.rstd, ln.gamma.f32, ln.dz, ln.gamma.g, ln.beta.g if ln.gamma.g is not None else None, dx_drop=ln.dzd,
                    drop_p=p, seed=ln.seed, defer=True)


def _ln_bwd_fused(ln, a, w, p, res=None, res2=None, fold=None, fold_cols=0, K=None):
    """gradient of ln's OUTPUT = a w (+ res + res2); the same launch applies ln's backward: ln.dz, ln.dzd"""
    M = a.shape[0]
    ln.dz = torch.empty(M, 256, dtype=BF16, device=a.device)
    ln.dzd = torch.empty(M, 256, dtype=BF16, device=a.device) if p > 0 else None
    k.rowgemm(a, w, ln.dz, b_kind=k.B_KROW, epi=k.ROW_LN_BWD, K=K, res=res, res2=res2, gamma=ln.gamma.f32, z=ln.z, mean=ln.mean, rstd=ln.rstd,
              out2=ln.dzd, drop_p=p, drop_seed=ln.seed, fold=fold, fold_cols=fold_cols, dgamma=ln.gamma.g,
              dbeta=ln.beta.g if ln.gamma.g is not None else None)


def _ffn_fwd(tape, x1, W1, b1):
    """dropout(relu(x1 W1^T + b1)) (transformer.py:301: linear2(dropout(activation(linear1(src)))))"""
    p = tape.drop_p
    seed = tape.next_seed() if p > 0 else 0
    kw = dict(drop_where=2, drop_p=p, drop_seed=seed) if p > 0 else {}
    return ops.linear(x1, W1.w, b1.f32, act=k.ACT_RELU, **kw)


def _ffn_bwd(tape, ln_out, ln_in, h, x_mid, W1, b1, W2, b2, p):
    """backward of ln_out = LN(x_mid + dropout(linear2(h))), h = dropout(relu(linear1(x_mid))), x_mid = ln_in's output:
    weight gradients, then [linear1 dgrad + residual gradient + ln_in's backward] as one launch"""
    gb = ln_out.dzd if p > 0 else ln_out.dz
    if W2.g is not None:
        tape.linear_wgrad(gb, h, W2, b2)
    dh = ops.linear_dgrad(gb, W2.w, act=k.ACT_MASK_POS, aux=h, alpha=1.0 / (1.0 - p) if p > 0 else 1.0)
    if W1.g is not None:
        tape.linear_wgrad(dh, x_mid, W1, b1)
    _ln_bwd_fused(ln_in, dh, W1.w, p, res=ln_out.dz)


def _outproj_bwd(tape, ln, ctx, Wo, bo, p):
    """gradient of the attention context from ln = LN(resid + dropout(ctx Wo^T + bo))"""
    go = ln.dzd if p > 0 else ln.dz
    if Wo.g is not None:
        tape.linear_wgrad(go, ctx, Wo, bo)
    return ops.linear_dgrad(go, Wo.w)


def _qkv(xe, x, Win, bin_, M, d):
    qkv = torch.empty(M, 3 * d, dtype=BF16, device=x.device)
    k.gemm(M, 3 * d, d, k.A_ROWK, k.operand(xe, xe.stride(0)), k.B_ROWK, k.operand(Win.w, Win.w.stride(0)), qkv, 3 * d, shift=bin_.f32,
           a2=x, a2_from=2 * d, flops=2 * M * 3 * d * d)
    return qkv


# ------------------------------------------------------------------------------------------------------------------ encoder
def encoder_program(tape, ps, x, pos, key_pad, B, S, H, n_layers):
    """6 post-norm encoder layers over batch-major tokens x [B*S, 256]; pos = bf16 constant of the same shape"""
    d, M, p = 256, B * S, tape.drop_p
    xe = torch.empty_like(x.data)
    k.add(x.data, pos, xe, b_period=pos.numel())
    cur, cur_e = x.data, xe
    layers = []
    for i in range(n_layers):
        lp = f"layers.{i}."
        L = SimpleNamespace(i=i, x_in=cur, xe_in=cur_e)
        L.Win, L.bin = ps[lp + "self_attn.in_proj_weight"], ps[lp + "self_attn.in_proj_bias"]
        L.Wo, L.bo = ps[lp + "self_attn.out_proj.weight"], ps[lp + "self_attn.out_proj.bias"]
        L.W1, L.b1, L.W2, L.b2 = ps[lp + "linear1.weight"], ps[lp + "linear1.bias"], ps[lp + "linear2.weight"], ps[lp + "linear2.bias"]
        qkv = _qkv(cur_e, cur, L.Win, L.bin, M, d)
        seed_p = tape.next_seed() if p > 0 else 0
        L.ctx = torch.empty(M, d, dtype=BF16, device=cur.device)
        L.core_bwd = _core(tape, qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], key_pad, B, S, S, H, L.ctx, p, seed_p)
        L.ln1 = _ln_fwd(L.ctx, L.Wo, L.bo, cur, ps[lp + "norm1.weight"], ps[lp + "norm1.bias"], tape)
        L.h = _ffn_fwd(tape, L.ln1.y, L.W1, L.b1)
        L.ln2 = _ln_fwd(L.h, L.W2, L.b2, L.ln1.y, ps[lp + "norm2.weight"], ps[lp + "norm2.bias"], tape, add=pos if i + 1 < n_layers else None)
        cur, cur_e = L.ln2.y, L.ln2.y2
        layers.append(L)
    out = engine.Var(cur)

    def make_bwd(L):
        def bwd():
            if L.ln2.dz is None:                       # last layer: the gradient of the program output
                g = out.take_grad()
                if g is None:
                    return
                _ln_bwd_alone(L.ln2, g, p)
            _ffn_bwd(tape, L.ln2, L.ln1, L.h, L.ln1.y, L.W1, L.b1, L.W2, L.b2, p)
            dctx = _outproj_bwd(tape, L.ln1, L.ctx, L.Wo, L.bo, p)
            dqkv = torch.empty(M, 3 * d, dtype=BF16, device=dctx.device)
            part = L.core_bwd(dctx, dqkv[:, :d], dqkv[:, d:2 * d], dqkv[:, 2 * d:])
            # d x = [dq | dk | dv] W_in + the gradient over the residual connection; below layer 0 it is the gradient of the program input.
            # The same launch folds the key-split shares of dq into dqkv[:, :d] (the weight gradients below read the folded rows).
            if L.i > 0:
                _ln_bwd_fused(layers[L.i - 1].ln2, dqkv, L.Win.w, p, res=L.ln1.dz, fold=part, fold_cols=d)
            elif x.needs_grad:
                gx = torch.empty(M, d, dtype=BF16, device=dctx.device)
                k.rowgemm(dqkv, L.Win.w, gx, b_kind=k.B_KROW, epi=k.ROW_PLAIN, res=L.ln1.dz, res2=x.grad, fold=part, fold_cols=d)
                x.grad = gx
            elif part is not None:
                dqkv[:, :d] = part.float().sum(0).to(BF16)
            if L.Win.g is not None:
                tape.linear_wgrad(dqkv[:, :2 * d], L.xe_in, L.Win.rows(0, 2 * d), L.bin.rows(0, 2 * d))
                tape.linear_wgrad(dqkv[:, 2 * d:], L.x_in, L.Win.rows(2 * d, 3 * d), L.bin.rows(2 * d, 3 * d))
            L.ln1.dz = L.ln1.dzd = L.ln2.dz = L.ln2.dzd = None
        return bwd

    for L in layers:
        tape.record(make_bwd(L))
    return [out], None


# ------------------------------------------------------------------------------------------------------------------ decoder
def _decoder_layers_xcd(tape, ps, Wself, Wcross, x0, qpos, kv, key_pad, tgt_stack, B, S, Q, H, L_):
    """Forward of all decoder layers as ONE launch (csrc/xdec.hip: one image per XCD, XCD-local barriers); returns the per-layer records the
    backward steps of decoder_program read -- every saved tensor has the layout / statistics / dropout hash of the per-op launches, so the
    backward pass is the same code.  Seeds are drawn in the order of the per-op path (tests compare the two paths WITH dropout)."""
    d, M, p = 256, B * Q, tape.drop_p
    dev = x0.device
    bf = lambda *shape: torch.empty(*shape, dtype=BF16, device=dev)
    f32 = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)
    out = dict(qkv=bf(L_, M, 3 * d), ctx_s=bf(L_, M, d), lse_s=f32(L_, B * H, Q, 2), z1=bf(L_, M, d), y1=bf(L_, M, d), y1e=bf(L_, M, d), mean1=f32(L_, M), rstd1=f32(L_, M),
               qc=bf(L_, M, d), ctx_c=bf(L_, M, d), lse_c=f32(L_, B * H, Q, 2), z3=bf(L_, M, d), y3=bf(L_, M, d), mean3=f32(L_, M), rstd3=f32(L_, M),
               h=bf(L_, M, 2048), z4=bf(L_, M, d), y4=tgt_stack, y4e=bf(L_, M, d), mean4=f32(L_, M), rstd4=f32(L_, M))
    part = bf(B * 32 * 128 * 256)
    recs, table = [], []
    for i in range(L_):
        lp = f"layers.{i}."
        L = SimpleNamespace(i=i, x_in=x0 if i == 0 else out["y4"][i - 1], xe_in=qpos if i == 0 else out["y4e"][i - 1])
        L.Ws, L.bs = Wself[i]
        L.Wc, L.bc = Wcross[i]
        L.Wq, L.bq = L.Wc.rows(0, d), L.bc.rows(0, d)
        L.Wos, L.bos = ps[lp + "self_attn.out_proj.weight"], ps[lp + "self_attn.out_proj.bias"]
        L.Woc, L.boc = ps[lp + "cross_attn_image.out_proj.weight"], ps[lp + "cross_attn_image.out_proj.bias"]
        L.W1, L.b1, L.W2, L.b2 = ps[lp + "linear1.weight"], ps[lp + "linear1.bias"], ps[lp + "linear2.weight"], ps[lp + "linear2.bias"]
        norms = [(ps[lp + f"norm{j}.weight"], ps[lp + f"norm{j}.bias"]) for j in (1, 3, 4)]
        seeds = [tape.next_seed() if p > 0 else 0 for _ in range(6)]      # self-attention, norm1, cross-attention, norm3, hidden, norm4
        qkv = out["qkv"][i]
        L.ctx_s, L.ctx_c, L.h, L.col = out["ctx_s"][i], out["ctx_c"][i], out["h"][i], i * 2 * d
        L.core_s = _core_bwd_of(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], None, B, Q, Q, H, L.ctx_s, p, seeds[0], out["lse_s"][i])
        L.core_c = _core_bwd_of(out["qc"][i], kv[:, L.col:L.col + d], kv[:, L.col + d:L.col + 2 * d], key_pad, B, Q, S, H, L.ctx_c, p, seeds[2], out["lse_c"][i])
        last = i + 1 == L_
        L.ln1 = SimpleNamespace(seed=seeds[1], z=out["z1"][i], y=out["y1"][i], y2=out["y1e"][i], mean=out["mean1"][i], rstd=out["rstd1"][i], gamma=norms[0][0],
                                beta=norms[0][1], dz=None, dzd=None)
        L.ln3 = SimpleNamespace(seed=seeds[3], z=out["z3"][i], y=out["y3"][i], y2=None, mean=out["mean3"][i], rstd=out["rstd3"][i], gamma=norms[1][0],
                                beta=norms[1][1], dz=None, dzd=None)
        L.ln4 = SimpleNamespace(seed=seeds[5], z=out["z4"][i], y=out["y4"][i], y2=None if last else out["y4e"][i], mean=out["mean4"][i], rstd=out["rstd4"][i],
                                gamma=norms[2][0], beta=norms[2][1], dz=None, dzd=None)
        L.out = engine.Var(L.ln4.y)
        table.append(dict(w_in=L.Ws.w, b_in=L.bs.f32, w_os=L.Wos.w, b_os=L.bos.f32, g1=norms[0][0].f32, be1=norms[0][1].f32, w_q=L.Wq.w, b_q=L.bq.f32, w_oc=L.Woc.w,
                          b_oc=L.boc.f32, g3=norms[1][0].f32, be3=norms[1][1].f32, w1=L.W1.w, b1=L.b1.f32, w2=L.W2.w, b2=L.b2.f32, g4=norms[2][0].f32,
                          be4=norms[2][1].f32, seed=seeds))
        recs.append(L)
    k.xdec_fwd(B, Q, S, x0, qpos, kv, key_pad, p, 1e-5, out, table, part)
    tape.keep.append((out, part))
    return recs, SimpleNamespace(out=out, part=part, table=table)


def _decoder_backward_xcd(tape, layers, fw, g_out, kv, dkv, sink, key_pad, B, S, Q, H, p):
    """Backward of all decoder layers: ONE launch for the data-gradient chain (csrc/xdec.hip xdec_bwd_kernel), then the weight gradients exactly as
    the per-op steps queue them (grouped launches at the program's end) and the LayerNorm parameter gradients as deferred folds of the
    launch's per-row-block partial sums."""
    d, M, L_ = 256, B * Q, len(layers)
    dev = kv.device
    bf = lambda *shape: torch.empty(*shape, dtype=BF16, device=dev)
    nblk = B * ((Q + 3) // 4)
    outs = dict(gb4=bf(L_, M, d), dh=bf(L_, M, 2048), go3=bf(L_, M, d), go1=bf(L_, M, d), sink=sink, dkv=dkv,
                ln_part=torch.empty(L_, 3, 2, nblk, d, dtype=torch.float32, device=dev))
    scratch = dict(dctx=bf(2, M, d), part=fw.part, dq_part=bf(4, M, d))
    table = [dict(w_in=L.Ws.w, w_os=L.Wos.w, w_q=L.Wq.w, w_oc=L.Woc.w, w1=L.W1.w, w2=L.W2.w, g1=L.ln1.gamma.f32, g3=L.ln3.gamma.f32, g4=L.ln4.gamma.f32,
                  seed=t["seed"]) for L, t in zip(layers, fw.table)]
    k.xdec_bwd(B, Q, S, kv, key_pad, p, fw.out, g_out, outs, table, scratch)
    for L in reversed(layers):
        i = L.i
        if L.W2.g is not None:
            tape.linear_wgrad(outs["gb4"][i], L.h, L.W2, L.b2)
        if L.W1.g is not None:
            tape.linear_wgrad(outs["dh"][i], L.ln3.y, L.W1, L.b1)
        if L.Woc.g is not None:
            tape.linear_wgrad(outs["go3"][i], L.ctx_c, L.Woc, L.boc)
        if L.Wq.g is not None:
            tape.linear_wgrad(sink[:, i * 4 * d + 3 * d:i * 4 * d + 4 * d], L.ln1.y2, L.Wq, L.bq)
        if L.Wos.g is not None:
            tape.linear_wgrad(outs["go1"][i], L.ctx_s, L.Wos, L.bos)
        if L.Ws.g is not None:
            dqkv = sink[:, i * 4 * d:i * 4 * d + 3 * d]
            tape.linear_wgrad(dqkv[:, :2 * d], L.xe_in, L.Ws.rows(0, 2 * d), L.bs.rows(0, 2 * d))
            tape.linear_wgrad(dqkv[:, 2 * d:], L.x_in, L.Ws.rows(2 * d, 3 * d), L.bs.rows(2 * d, 3 * d))
        for which, ln in ((0, L.ln1), (1, L.ln3), (2, L.ln4)):
            if ln.gamma.g is not None:
                k.queue_fold(outs["ln_part"][i, which, 0], ln.gamma.g, nblk, keep=(outs["ln_part"],))
                k.queue_fold(outs["ln_part"][i, which, 1], ln.beta.g, nblk, keep=(outs["ln_part"],))
    tape.keep.append((outs, scratch))


def decoder_program(tape, ps, mem, qe, pos, key_pad, B, S, Q, H, n_layers):
    """6 decoder layers (self-attention over the queries, cross-attention into the encoder memory, FFN) + the shared final LayerNorm over
    all layer outputs; returns [hs] with hs [L, B*Q, 256] bf16 (transformer.py:225-267, 362-408)."""
    d, M, p, L_ = 256, B * Q, tape.drop_p, n_layers
    dev = mem.data.device
    qpos = torch.empty(B, Q, d, dtype=BF16, device=dev)
    qpos.copy_(qe.data.unsqueeze(0).expand(B, Q, d))          # cast + broadcast over the batch in one launch
    qpos = qpos.view(M, d)
    Wself = [(ps[f"layers.{i}.self_attn.in_proj_weight"], ps[f"layers.{i}.self_attn.in_proj_bias"]) for i in range(L_)]
    Wcross = [(ps[f"layers.{i}.cross_attn_image.in_proj_weight"], ps[f"layers.{i}.cross_attn_image.in_proj_bias"]) for i in range(L_)]
    need = qe.needs_grad or mem.needs_grad or Wself[0][0].g is not None
    sink = torch.empty(M, L_ * 4 * d, dtype=BF16, device=dev) if need else None       # per layer [dq_s | dk_s | dv_s | dq_c]

    def qpos_bwd():     # recorded first: runs after every layer has written its slice of `sink`
        if not qe.needs_grad or sink is None:
            return
        zero = torch.zeros(d, d, dtype=BF16, device=dev)
        wst = torch.cat([t for i in range(L_) for t in (Wself[i][0].w[:2 * d], zero, Wcross[i][0].w[:d])], dim=0)     # [L*4d, d]
        gq = ops.linear_dgrad(sink, wst).view(B, Q, d).float().sum(0)
        qe.grad = gq if qe.grad is None else qe.grad + gq

    tape.record(qpos_bwd)
    mem_e = torch.empty_like(mem.data)
    k.add(mem.data, pos, mem_e, b_period=pos.numel())
    kv, dkv = engine.cross_kv_projections(tape, mem, mem_e, Wcross)
    tgt_stack = torch.empty(L_, M, d, dtype=BF16, device=dev)
    cur = torch.zeros(M, d, dtype=BF16, device=dev)
    cur_e = qpos
    layers = []
    ff = ps["layers.0.linear1.weight"].w.shape[0]
    fused = XDEC and all(ps[f"layers.{i}.linear1.weight"].w.shape[0] == ff for i in range(L_)) and k.xdec_supported(B, Q, S, L_, ff=ff)
    fw = None
    if fused:
        layers, fw = _decoder_layers_xcd(tape, ps, Wself, Wcross, cur, qpos, kv, key_pad, tgt_stack, B, S, Q, H, L_)
    for i in range(0 if not fused else L_, L_):
        lp = f"layers.{i}."
        L = SimpleNamespace(i=i, x_in=cur, xe_in=cur_e)
        L.Ws, L.bs = Wself[i]
        L.Wc, L.bc = Wcross[i]
        L.Wq, L.bq = L.Wc.rows(0, d), L.bc.rows(0, d)
        L.Wos, L.bos = ps[lp + "self_attn.out_proj.weight"], ps[lp + "self_attn.out_proj.bias"]
        L.Woc, L.boc = ps[lp + "cross_attn_image.out_proj.weight"], ps[lp + "cross_attn_image.out_proj.bias"]
        L.W1, L.b1, L.W2, L.b2 = ps[lp + "linear1.weight"], ps[lp + "linear1.bias"], ps[lp + "linear2.weight"], ps[lp + "linear2.bias"]
        # self-attention: q = k = tgt + query_pos, v = tgt
        qkv = _qkv(cur_e, cur, L.Ws, L.bs, M, d)
        seed_s = tape.next_seed() if p > 0 else 0
        L.ctx_s = torch.empty(M, d, dtype=BF16, device=dev)
        L.core_s = _core(tape, qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], None, B, Q, Q, H, L.ctx_s, p, seed_s)
        L.ln1 = _ln_fwd(L.ctx_s, L.Wos, L.bos, cur, ps[lp + "norm1.weight"], ps[lp + "norm1.bias"], tape, add=qpos)
        # cross-attention: q = t1 + query_pos, k = memory + pos, v = memory (k, v projected for all layers up front)
        qc = ops.linear(L.ln1.y2, L.Wq.w, L.bq.f32)
        seed_c = tape.next_seed() if p > 0 else 0
        L.ctx_c = torch.empty(M, d, dtype=BF16, device=dev)
        col = i * 2 * d
        L.col = col
        L.core_c = _core(tape, qc, kv[:, col:col + d], kv[:, col + d:col + 2 * d], key_pad, B, Q, S, H, L.ctx_c, p, seed_c)
        L.ln3 = _ln_fwd(L.ctx_c, L.Woc, L.boc, L.ln1.y, ps[lp + "norm3.weight"], ps[lp + "norm3.bias"], tape)
        L.h = _ffn_fwd(tape, L.ln3.y, L.W1, L.b1)
        L.ln4 = _ln_fwd(L.h, L.W2, L.b2, L.ln3.y, ps[lp + "norm4.weight"], ps[lp + "norm4.bias"], tape, add=qpos if i + 1 < L_ else None,
                        y=tgt_stack[i])
        L.out = engine.Var(L.ln4.y)          # receives the gradient of the shared final norm (split_bwd below)
        cur, cur_e = L.ln4.y, L.ln4.y2
        layers.append(L)

    def make_bwd(L):
        def bwd():
            i = L.i
            if L.ln4.dz is None:                      # last layer: only the shared final norm consumes its output
                g = L.out.take_grad()
                if g is None:
                    return
                _ln_bwd_alone(L.ln4, g, p)
            _ffn_bwd(tape, L.ln4, L.ln3, L.h, L.ln3.y, L.W1, L.b1, L.W2, L.b2, p)
            # cross-attention
            dctx = _outproj_bwd(tape, L.ln3, L.ctx_c, L.Woc, L.boc, p)
            dq = sink[:, i * 4 * d + 3 * d:i * 4 * d + 4 * d] if sink is not None else torch.empty(M, d, dtype=BF16, device=dev)
            part = L.core_c(dctx, dq, dkv[:, L.col:L.col + d], dkv[:, L.col + d:L.col + 2 * d])
            _ln_bwd_fused(L.ln1, dq, L.Wq.w, p, res=L.ln3.dz, K=d, fold=part, fold_cols=d)       # also folds the key-split shares of dq
            if L.Wq.g is not None:
                tape.linear_wgrad(dq, L.ln1.y2, L.Wq, L.bq)
            # self-attention
            dctx = _outproj_bwd(tape, L.ln1, L.ctx_s, L.Wos, L.bos, p)
            dqkv = sink[:, i * 4 * d:i * 4 * d + 3 * d] if sink is not None else torch.empty(M, 3 * d, dtype=BF16, device=dev)
            part = L.core_s(dctx, dqkv[:, :d], dqkv[:, d:2 * d], dqkv[:, 2 * d:])
            if i > 0:
                # gradient of the layer below's output: in_proj dgrad + residual gradient + its share of the final norm's gradient
                below = layers[i - 1]
                _ln_bwd_fused(below.ln4, dqkv, L.Ws.w, p, res=L.ln1.dz, res2=below.out.take_grad(), K=3 * d, fold=part, fold_cols=d)
            elif part is not None:
                dqkv[:, :d] = part.float().sum(0).to(BF16)
            if L.Ws.g is not None:
                tape.linear_wgrad(dqkv[:, :2 * d], L.xe_in, L.Ws.rows(0, 2 * d), L.bs.rows(0, 2 * d))
                tape.linear_wgrad(dqkv[:, 2 * d:], L.x_in, L.Ws.rows(2 * d, 3 * d), L.bs.rows(2 * d, 3 * d))
            L.ln1.dz = L.ln1.dzd = L.ln3.dz = L.ln3.dzd = L.ln4.dz = L.ln4.dzd = None
        return bwd

    # one launch for the whole data-gradient chain when the forward ran as one launch, every decoder weight trains and dk / dv have a consumer
    fused_bwd = fused and XDEC_BWD and sink is not None and dkv is not None and all(
        v.g is not None for L in layers for v in (L.Ws, L.Wos, L.Wq, L.Woc, L.W1, L.W2, L.ln1.gamma, L.ln3.gamma, L.ln4.gamma))
    shared = SimpleNamespace(g=None)
    if fused_bwd:
        def all_layers_bwd():
            if shared.g is not None:
                _decoder_backward_xcd(tape, layers, fw, shared.g, kv, dkv, sink, key_pad, B, S, Q, H, p)
                shared.g = None
        tape.record(all_layers_bwd)
    else:
        for L in layers:
            tape.record(make_bwd(L))
    allv = engine.Var(tgt_stack.view(L_ * M, d))

    def split_bwd():    # gradient of the shared final norm -> the layer outputs (runs before the layers' own backward steps)
        g = allv.take_grad()
        if g is None:
            return
        g = g.view(L_, M, d)
        if fused_bwd:
            shared.g = g.contiguous()
            return
        for i, L in enumerate(layers):
            L.out.grad = g[i]

    tape.record(split_bwd)
    stack = torch.empty(L_, M, d, dtype=BF16, device=dev)
    hs_flat = engine.layernorm(tape, allv, ps["norm.weight"], ps["norm.bias"], 1e-5, y=stack.view(L_ * M, d))
    hs = engine.Var(stack)

    def hs_bwd():
        g = hs.take_grad()
        if g is not None:
            hs_flat.grad = g.reshape(L_ * M, d)

    tape.record(hs_bwd)
    return [hs], None
