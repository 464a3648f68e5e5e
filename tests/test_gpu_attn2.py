"""Second-generation attention cores (csrc/attn2.hip: flash-style forward with a key-block loop, key-owning backward) against fp32
autograd of softmax(scale q k^T + key padding) -> dropout -> P v (nn.MultiheadAttention's core, /root/reference/models/transformer.py:297,
370-400) on the same bf16-rounded operands.  The dropout mask is the kernels' own hash, re-implemented here bit for bit in integer
torch ops, so that forward AND backward are held against autograd WITH dropout.  Shapes: the benchmark's (416 x 416, 100 x 416,
100 x 100), ragged ones, and key counts beyond the first generation's 480-key limit (800 x 1333-pixel inputs give ~1100 tokens)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

BF = torch.bfloat16


def pair_hash(pair, seed):
    """csrc/attn2.hip pair_hash in int64 arithmetic (values kept below 2^32)"""
    M32 = 0xFFFFFFFF
    s0 = seed & M32
    s1 = ((seed >> 32) ^ ((seed & M32) * 0x9E3779B9)) & M32
    a = (pair ^ s0) & M32
    a = a ^ (a >> 12)
    h = ((a & 0xFFFFFF) * 0x9E3779 + s1) & M32
    h = h ^ (h >> 15)
    h = ((h & 0xFFFFFF) * 0x85EBCB + (a >> 8)) & M32
    h = h ^ (h >> 13)
    return h


def keep_mask(BH, Sq, Sk, p, seed):
    ldp = (Sk + 7) // 8 * 8
    row = torch.arange(BH * Sq, dtype=torch.int64).view(BH, Sq, 1)
    key = torch.arange(Sk, dtype=torch.int64).view(1, 1, Sk)
    idx = row * ldp + key
    h = pair_hash(idx >> 1, seed)
    field = torch.where((key & 1) == 1, h >> 16, h & 0xFFFF)
    return field >= int(p * 65536.0 + 0.5)


@pytest.mark.parametrize("B,Sq,Sk,drop,pad", [(8, 416, 416, 0.1, False), (8, 100, 416, 0.1, True), (8, 100, 100, 0.1, False), (2, 70, 530, 0.0, True),
                                             (2, 1100, 1100, 0.1, True), (3, 33, 17, 0.25, True), (2, 130, 513, 0.1, False)])
def test_attn2_forward_backward_against_autograd(dev, B, Sq, Sk, drop, pad):
    from toist_amd import kernels as k
    H, dh, d = 8, 32, 256
    g = torch.Generator().manual_seed(Sq * 7 + Sk)
    q = (torch.randn(B * Sq, d, generator=g) * 1.5).to(BF)
    kk = (torch.randn(B * Sk, d, generator=g) * 1.5).to(BF)
    v = torch.randn(B * Sk, d, generator=g).to(BF)
    dctx = torch.randn(B * Sq, d, generator=g).to(BF)
    key_pad = None
    if pad:
        key_pad = torch.zeros(B, Sk, dtype=torch.uint8)
        for b in range(B):
            n = 1 + (b * 37) % max(1, Sk // 3)
            key_pad[b, Sk - n:] = 1                                   # padded tail (text padding)
            key_pad[b, (b * 11) % max(1, Sk - n)] = 1                 # and one masked key in the middle
    scale = 1.0 / math.sqrt(dh)
    seed = 0x1234567 + Sq
    # a packed [q | k | v]-like layout: column slices with a row stride of 3 d
    qd = torch.empty(B * Sq, 3 * d, dtype=BF, device=dev)
    kvd = torch.empty(B * Sk, 3 * d, dtype=BF, device=dev)
    qd[:, :d], kvd[:, d:2 * d], kvd[:, 2 * d:] = q.to(dev), kk.to(dev), v.to(dev)
    ctx = torch.empty(B * Sq, d, dtype=BF, device=dev)
    lse = torch.empty(B * H, Sq, 2, device=dev)
    kp_dev = key_pad.to(dev) if key_pad is not None else None
    k.SEED_DEV = None
    k.attn2_fwd(qd[:, :d], kvd[:, d:2 * d], kvd[:, 2 * d:], kp_dev, B, H, Sq, Sk, dh, scale, drop, seed, ctx, lse)

    # ---- fp32 reference with the same keep mask ----
    qr = q.float().view(B, Sq, H, dh).permute(0, 2, 1, 3).reshape(B * H, Sq, dh).requires_grad_(True)
    kr = kk.float().view(B, Sk, H, dh).permute(0, 2, 1, 3).reshape(B * H, Sk, dh).requires_grad_(True)
    vr = v.float().view(B, Sk, H, dh).permute(0, 2, 1, 3).reshape(B * H, Sk, dh).requires_grad_(True)
    raw = qr @ kr.transpose(1, 2)
    s = raw * scale
    if key_pad is not None:
        s = s.masked_fill(key_pad.bool().view(B, 1, 1, Sk).expand(B, H, Sq, Sk).reshape(B * H, Sq, Sk), float("-inf"))
    P = s.softmax(-1)
    if drop > 0:
        keep = keep_mask(B * H, Sq, Sk, drop, seed)
        frac = 1.0 - float(keep.float().mean())
        assert abs(frac - drop) < 0.01, frac
        Pd = P * keep / (1.0 - drop)
    else:
        Pd = P
    out = Pd @ vr
    ref_ctx = out.view(B, H, Sq, dh).permute(0, 2, 1, 3).reshape(B * Sq, d)
    err = (ctx.float().cpu() - ref_ctx.detach()).abs()
    bound = 1.2e-2 * ref_ctx.detach().abs() + 6e-3          # P and the context are bf16: 2^-8 relative each
    assert bool((err <= bound).all()), f"context: max excess {float((err - bound).max()):.3e}"
    # row statistics: maximum of the raw dot products (over live keys), 1 / sum of exp(scale (s - max))
    live = torch.ones(B * H, 1, Sk, dtype=torch.bool) if key_pad is None else ~key_pad.bool().view(B, 1, 1, Sk).expand(B, H, 1, Sk).reshape(B * H, 1, Sk)
    rmax = raw.detach().masked_fill(~live, float("-inf")).amax(-1)
    torch.testing.assert_close(lse[..., 0].cpu(), rmax, rtol=1e-5, atol=1e-4)
    rsum = (torch.exp((raw.detach() - rmax[..., None]) * scale) * live).sum(-1)
    torch.testing.assert_close(lse[..., 1].cpu(), 1.0 / rsum, rtol=2e-3, atol=1e-6)

    # ---- backward ----
    out.backward(dctx.float().view(B, Sq, H, dh).permute(0, 2, 1, 3).reshape(B * H, Sq, dh))
    splits = k.attn2_splits(Sk)
    assert splits == (((Sk + 31) // 32) + 3) // 4
    dq = torch.full((B * Sq, d), float("nan"), dtype=BF, device=dev)
    dkv = torch.full((B * Sk, 2 * d), float("nan"), dtype=BF, device=dev)
    part = torch.full((splits, B * Sq, d), float("nan"), dtype=BF, device=dev) if splits > 1 else None
    k.attn2_bwd(qd[:, :d], kvd[:, d:2 * d], kvd[:, 2 * d:], ctx, dctx.to(dev), lse, kp_dev, B, H, Sq, Sk, dh, scale, drop, seed,
                dq if splits == 1 else None, dkv[:, :d], dkv[:, d:], dq_part=part)
    got_dq = dq.float().cpu() if splits == 1 else part.float().sum(0).cpu()
    back = lambda t, S: t.view(B, H, S, dh).permute(0, 2, 1, 3).reshape(B * S, d)
    for name, got, ref in (("dq", got_dq, back(qr.grad, Sq)), ("dk", dkv[:, :d].float().cpu(), back(kr.grad, Sk)), ("dv", dkv[:, d:].float().cpu(), back(vr.grad, Sk))):
        assert bool(torch.isfinite(got).all()), f"{name}: not every element was written"
        rel = float((got - ref).norm() / (ref.norm() + 1e-20))
        assert rel < 1.5e-2, f"{name}: relative Frobenius error {rel:.4f}"
        err = (got - ref).abs()
        bound = 3e-2 * ref.abs() + 3e-2 * float(ref.abs().mean()) + 1e-4
        assert float((err > bound).float().mean()) < 1e-3, f"{name}: {float((err > bound).float().mean()):.4f} of the elements off"


def test_attn2_large_scores_and_seed_word(dev):
    """dot products of 1e11 (an undamped random-init backbone; the second training step of the bench model produced NaN with a fused
    s c - m c): the kernels subtract the row maximum first -- exactly -- and scale the difference, forward and backward re-form the same
    probabilities; the device seed word shifts the dropout mask of a replayed graph."""
    from toist_amd import kernels as k
    B, H, Sq, Sk, dh, d = 2, 8, 64, 160, 32, 256
    g = torch.Generator().manual_seed(1)
    q = (torch.randn(B * Sq, d, generator=g) * 2e5).to(BF).to(dev)
    kk = (torch.randn(B * Sk, d, generator=g) * 2e5).to(BF).to(dev)
    v = torch.randn(B * Sk, d, generator=g).to(BF).to(dev)
    scale = 1.0 / math.sqrt(dh)
    ctx, lse = torch.empty(B * Sq, d, dtype=BF, device=dev), torch.empty(B * H, Sq, 2, device=dev)
    k.SEED_DEV = None
    k.attn2_fwd(q, kk, v, None, B, H, Sq, Sk, dh, scale, 0.0, 0, ctx, lse)
    qr = q.float().cpu().view(B, Sq, H, dh).permute(0, 2, 1, 3)
    kr = kk.float().cpu().view(B, Sk, H, dh).permute(0, 2, 1, 3)
    vr = v.float().cpu().view(B, Sk, H, dh).permute(0, 2, 1, 3)
    P = (qr @ kr.transpose(-1, -2) * scale).softmax(-1)               # one-hot rows
    ref = (P @ vr).permute(0, 2, 1, 3).reshape(B * Sq, d)
    assert bool(torch.isfinite(ctx).all()) and bool(torch.isfinite(lse).all())
    off = (ctx.float().cpu() - ref).abs().amax(1) > 6e-2
    assert int(off.sum()) <= 2, f"{int(off.sum())} of {B * Sq} one-hot rows picked another winner"
    # the backward pass on the same operands: finite everywhere, dv = P^T dctx of one-hot rows
    dctx = torch.randn(B * Sq, d, generator=g).to(BF).to(dev)
    dk, dv = (torch.empty(B * Sk, d, dtype=BF, device=dev) for _ in range(2))
    part = torch.empty(k.attn2_splits(Sk), B * Sq, d, dtype=BF, device=dev)
    k.attn2_bwd(q, kk, v, ctx, dctx, lse, None, B, H, Sq, Sk, dh, scale, 0.0, 0, None, dk, dv, dq_part=part)
    assert bool(torch.isfinite(part).all()) and bool(torch.isfinite(dk).all()) and bool(torch.isfinite(dv).all())
    dv_ref = (P.transpose(-1, -2) @ dctx.float().cpu().view(B, Sq, H, dh).permute(0, 2, 1, 3)).permute(0, 2, 1, 3).reshape(B * Sk, d)
    assert float((dv.float().cpu() - dv_ref).norm() / dv_ref.norm()) < 5e-2
    # the device seed word: same host seed, different word -> a different mask; same word -> the same context bit for bit
    c1, c2, c3 = (torch.empty_like(ctx) for _ in range(3))
    try:
        k.SEED_DEV = torch.zeros(1, dtype=torch.int64, device=dev)
        k.attn2_fwd(q, kk, v, None, B, H, Sq, Sk, dh, scale, 0.3, 5, c1, lse)
        k.attn2_fwd(q, kk, v, None, B, H, Sq, Sk, dh, scale, 0.3, 5, c2, lse)
        k.SEED_DEV.add_(1000003)
        k.attn2_fwd(q, kk, v, None, B, H, Sq, Sk, dh, scale, 0.3, 5, c3, lse)
    finally:
        k.SEED_DEV = None
    assert torch.equal(c1, c2) and not torch.equal(c1, c3)
