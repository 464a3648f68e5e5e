"""GPU parity of the bf16 MFMA GEMM / implicit-conv family against fp32 CPU math on the same
bf16-rounded inputs.  Tolerance: the output is bf16 (rel 2^-8) with fp32 accumulation, so
|err| <= 1e-2 * |ref| + 2e-2 * sqrt(K)-scaled absolute slack (stated per test)."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

BF = torch.bfloat16


def _rand(shape, gen, scale=1.0):
    return (torch.randn(*shape, generator=gen) * scale).to(BF)


def _close(got, ref, K, name, rtol=1.5e-2, atol_unit=4e-3):
    got = got.float().cpu()
    atol = atol_unit * math.sqrt(K)
    err = (got - ref).abs()
    bad = err > (atol + rtol * ref.abs())
    assert not bool(bad.any()), f"{name}: {int(bad.sum())}/{bad.numel()} off, max err {float(err.max()):.4g} (atol {atol:.3g})"


@pytest.mark.parametrize("M,N,K,tile", [(128, 128, 32, 128), (200, 72, 96, 64), (333, 256, 256, 0), (100, 4, 256, 64), (1024, 512, 2048, 128)])
def test_linear_fwd(dev, M, N, K, tile):
    from toist_amd import kernels as k, ops
    g = torch.Generator().manual_seed(M + N + K)
    x, w = _rand((M, K), g), _rand((N, K), g, 0.1)
    bias = torch.randn(N, generator=g)
    ref = x.float() @ w.float().t() + bias
    out = ops.linear(x.to(dev), w.to(dev), bias.to(dev), tile=tile)
    _close(out, ref, K, "linear")
    out32 = ops.linear(x.to(dev), w.to(dev), bias.to(dev), out_dtype=torch.float32, tile=tile)
    _close(out32, ref, K, "linear f32 out", rtol=2e-3, atol_unit=1e-4)


def test_linear_epilogues(dev):
    from toist_amd import kernels as k, ops
    g = torch.Generator().manual_seed(7)
    M, N, K = 300, 264, 128
    x, w = _rand((M, K), g), _rand((N, K), g, 0.1)
    bias, scale = torch.randn(N, generator=g), torch.rand(N, generator=g) + 0.5
    res = _rand((M, N), g)
    base = (x.float() @ w.float().t()) * scale + bias + res.float()
    xd, wd, bd, sd, rd = x.to(dev), w.to(dev), bias.to(dev), scale.to(dev), res.to(dev)
    pre = torch.empty(M, N, dtype=BF, device=dev)
    out = ops.linear(xd, wd, bd, scale=sd, res=rd, act=k.ACT_RELU, pre_out=pre)
    _close(out, base.clamp(min=0), K, "relu")
    _close(pre, base, K, "pre_out")
    out = ops.linear(xd, wd, bd, scale=sd, res=rd, act=k.ACT_GELU)
    _close(out, F.gelu(base), K, "gelu")
    out = ops.linear(xd, wd, bd, scale=sd, res=rd, act=k.ACT_SIGMOID)
    _close(out, torch.sigmoid(base), K, "sigmoid")


@pytest.mark.parametrize("flags", [0, 1])
def test_linear_dgrad_wgrad(dev, flags):
    from toist_amd import kernels as k, ops
    g = torch.Generator().manual_seed(11)
    M, N, K = 416, 264, 136
    dy, w, x = _rand((M, N), g), _rand((N, K), g, 0.1), _rand((M, K), g)
    aux = _rand((M, K), g)
    dx_ref = dy.float() @ w.float()
    dx = ops.linear_dgrad(dy.to(dev), w.to(dev), flags=flags)
    _close(dx, dx_ref, N, f"dgrad flags={flags}")
    dxm = ops.linear_dgrad(dy.to(dev), w.to(dev), act=k.ACT_MASK_POS, aux=aux.to(dev), alpha=2.0, flags=flags)
    _close(dxm, torch.where(aux.float() > 0, 2.0 * dx_ref, torch.zeros_like(dx_ref)), N, "dgrad mask")
    dw_ref = dy.float().t() @ x.float()
    for sk in (1, 4):
        dw = ops.linear_wgrad(dy.to(dev), x.to(dev), flags=flags, split_k=sk)
        _close(dw, dw_ref, M, f"wgrad flags={flags} split={sk}", rtol=3e-3, atol_unit=1e-4)
    db = ops.bias_grad(dy.to(dev))
    _close(db, dy.float().sum(0), M, "bias grad", rtol=3e-3, atol_unit=1e-4)


@pytest.mark.parametrize("M,N,K", [(128, 768, 3072), (100, 256, 2048), (700, 260, 2304)])
def test_split_k_with_the_complete_epilogue(dev, M, N, K):
    """TOIST_GEMM_SPLIT_EPILOGUE (ops.linear on few tiles / deep K): partial tiles + splitk_epilogue_kernel must give what the un-split
    launch gives -- bias, dropout (the same mask: it is a function of (seed, row, column)), residual, GELU, bf16 output; run to run
    bit-identical (the slices are added in slice order)."""
    from toist_amd import kernels as k, ops
    g = torch.Generator().manual_seed(M + N)
    x = torch.randn(M, K, generator=g).to(BF).to(dev)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(BF).to(dev)
    bias, res = torch.randn(N, generator=g).to(dev), torch.randn(M, N, generator=g).to(BF).to(dev)
    k.SEED_DEV = torch.zeros(1, dtype=torch.int64, device=dev)
    assert ops._split_for_linear(M, N, K) > 1
    kw = dict(res=res, act=k.ACT_GELU, drop_where=1, drop_p=0.1, drop_seed=7)
    one = ops.linear(x, w, bias, split_k=1, **kw).float()
    for s in (None, 3, 16):
        got = ops.linear(x, w, bias, split_k=s, **kw).float()
        again = ops.linear(x, w, bias, split_k=s, **kw).float()
        assert torch.equal(got, again)
        assert float((got - one).abs().max()) <= 2.0 ** -6 * float(one.abs().max())
        assert float(((got == 0) != (one == 0)).float().mean()) < 1e-3          # same dropout mask (up to values rounding to zero)
    if N % 8:
        return                                                                  # k-major weights need N % 8 == 0
    wt = w.t().contiguous()                                                     # data gradient: k-major weights, ReLU mask from aux
    aux = torch.randn(M, N, generator=g).to(BF).to(dev)
    one = ops.linear_dgrad(x, wt, res=res, act=k.ACT_MASK_POS, aux=aux, split_k=1).float()
    got = ops.linear_dgrad(x, wt, res=res, act=k.ACT_MASK_POS, aux=aux).float()
    assert float((got - one).abs().max()) <= 2.0 ** -6 * float(one.abs().max())
    ref = torch.where(aux.float() > 0, x.float() @ wt.float() + res.float(), torch.zeros((), device=dev))
    _close(got.to(BF), ref.cpu(), K, "split dgrad")


def _nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize("C,Co,R,stride,pad,H,W", [(64, 64, 3, 1, 1, 20, 24), (64, 128, 3, 2, 1, 21, 20), (256, 512, 1, 2, 0, 16, 16),
                                                   (128, 256, 1, 1, 0, 10, 12), (8, 64, 7, 2, 3, 40, 40),
                                                   # 3x3 / stride 1 at >= 2048 pixels: the shared-halo kernel (conv3_kernel), forward and dgrad
                                                   (128, 192, 3, 1, 1, 40, 40), (64, 72, 3, 1, 1, 31, 23), (256, 64, 3, 1, 1, 46, 46),
                                                   # mask-head shapes: <= 32 output channels -> the narrow 128x32 / 32x128 tiles
                                                   (32, 16, 3, 1, 1, 48, 64), (64, 32, 3, 1, 1, 40, 40)])
def test_conv_fwd_bwd(dev, C, Co, R, stride, pad, H, W):
    from toist_amd import kernels as k, ops
    g = torch.Generator().manual_seed(C + Co + R)
    Nb = 3
    x = torch.randn(Nb, C, H, W, generator=g).to(BF)
    w = (torch.randn(Co, C, R, R, generator=g) * (1.0 / math.sqrt(C * R * R))).to(BF)
    scale, shift = torch.rand(Co, generator=g) + 0.5, torch.randn(Co, generator=g)
    xr = x.float().requires_grad_(True)
    wr = w.float().requires_grad_(True)
    y = F.conv2d(xr, wr, stride=stride, padding=pad)
    OH, OW = y.shape[-2:]
    res = torch.randn(Nb, Co, OH, OW, generator=g).to(BF)
    ref = F.relu(y * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1) + res.float())
    x_d, w_d = _nhwc(x).to(dev), _nhwc(w).to(dev)  # NHWC / KRSC
    out = ops.conv2d(x_d, w_d, stride=stride, pad=pad, scale=scale.to(dev), shift=shift.to(dev), res=_nhwc(res).to(dev), act=k.ACT_RELU)
    _close(out, _nhwc(ref.detach()), C * R * R, "conv fwd")
    if C == 8:
        return  # stem is frozen: no backward needed
    dy = torch.randn(Nb, Co, OH, OW, generator=g).to(BF)
    y.backward(dy.float())
    for flags in (0, 1):
        dx = ops.conv2d_dgrad(_nhwc(dy).to(dev), w_d, (H, W), stride=stride, pad=pad, flags=flags)
        if R == 1 and stride > 1:
            # scatter form: untouched rows keep their previous content -> compare only written rows
            full = torch.zeros(Nb, H, W, C, dtype=BF, device=dev)
            dx = ops.conv2d_dgrad(_nhwc(dy).to(dev), w_d, (H, W), stride=stride, pad=pad, out=full, flags=flags)
        _close(dx, _nhwc(xr.grad), Co * R * R, f"conv dgrad flags={flags}")
        dw = ops.conv2d_wgrad(_nhwc(dy).to(dev), x_d, (Co, R, R, C), stride=stride, pad=pad, flags=flags)
        _close(dw, _nhwc(wr.grad), Nb * OH * OW, f"conv wgrad flags={flags}", rtol=3e-3, atol_unit=2e-4)


@pytest.mark.parametrize("C,Co,H,W", [(64, 128, 21, 20), (128, 64, 16, 33), (64, 64, 1, 9), (64, 64, 7, 1), (256, 256, 40, 40)])
def test_conv3x3_stride2_dgrad_parity_classes(dev, C, Co, H, W):
    """ops._dgrad3x3_s2 (four stride-1 gathers, one per parity of the dx pixel) against autograd and against the generic
    transposed gather it replaces, with the fused residual + ReLU mask of the bottleneck backward; odd and degenerate planes."""
    from toist_amd import kernels as k, ops
    g = torch.Generator().manual_seed(C + H + W)
    Nb = 2
    x = torch.randn(Nb, C, H, W, generator=g).float().requires_grad_(True)
    w = (torch.randn(Co, C, 3, 3, generator=g) * (1.0 / math.sqrt(C * 9))).to(BF)
    y = F.conv2d(x, w.float(), stride=2, padding=1)
    dy = torch.randn(*y.shape, generator=g).to(BF)
    y.backward(dy.float())
    aux = torch.randn(Nb, H, W, C, generator=g).to(BF)
    res = torch.randn(Nb, H, W, C, generator=g).to(BF)
    ref = torch.where(aux.float() > 0, _nhwc(x.grad) + res.float(), torch.zeros(()))
    args = (_nhwc(dy).to(dev), _nhwc(w).to(dev), (H, W))
    kw = dict(stride=2, pad=1, res=res.to(dev), act=k.ACT_MASK_POS, aux=aux.to(dev))
    assert ops.PARITY_DGRAD
    got = ops.conv2d_dgrad(*args, **kw)
    _close(got, ref, Co * 9, "stride-2 dgrad by parity classes")
    ops.PARITY_DGRAD = False
    try:
        old = ops.conv2d_dgrad(*args, **kw)
    finally:
        ops.PARITY_DGRAD = True
    assert float((got.float() - old.float()).abs().max()) <= 2.0 ** -6 * float(old.float().abs().max())


def test_conv3_halo_kernel_is_used_and_agrees_with_generic(dev):
    """tile code 131 forces the shared-halo 3x3 kernel (error if it does not apply); it must agree with the generic
    implicit-GEMM tiles to bf16 rounding (fp32 sums in a different order) on a ResNet layer3-shaped problem."""
    from toist_amd import kernels as k, ops
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 40, 40, 256, generator=g).to(BF).to(dev)
    w = (torch.randn(256, 3, 3, 256, generator=g) * 0.02).to(BF).to(dev)
    shift = torch.randn(256, generator=g).to(dev)
    a = ops.conv2d(x, w, stride=1, pad=1, shift=shift, act=k.ACT_RELU, tile=131).float()
    b = ops.conv2d(x, w, stride=1, pad=1, shift=shift, act=k.ACT_RELU, tile=65).float()
    assert float((a - b).abs().max()) <= 2.0 ** -7 * float(b.abs().max())
    assert float((a - b).abs().mean()) <= 1e-3 * float(b.abs().mean())
    with pytest.raises(RuntimeError):   # stride 2 is not covered
        ops.conv2d(x, w, stride=2, pad=1, tile=131)


@pytest.mark.parametrize("C,Co", [(32, 16), (16, 8), (32, 32), (8, 16)])
def test_small_channel_conv_direct_kernel(dev, C, Co):
    """csrc/smallconv.hip (3x3 / s1 / p1, <= 32 channels, the 160x160 mask-head stages): forward with bias (+ residual) and the
    data gradient against fp32 conv2d on the same bf16-rounded inputs; >= 65536 pixels so ops.conv2d dispatches to it."""
    from toist_amd import kernels as k, ops
    g = torch.Generator().manual_seed(C * 100 + Co)
    Nb, H, W = 3, 150, 152
    x = torch.randn(Nb, C, H, W, generator=g).to(BF)
    w = (torch.randn(Co, C, 3, 3, generator=g) * (1.0 / math.sqrt(C * 9))).to(BF)
    bias = torch.randn(Co, generator=g)
    res = torch.randn(Nb, Co, H, W, generator=g).to(BF)
    xr, wr = x.float().requires_grad_(True), w.float()
    y = F.conv2d(xr, wr, padding=1)
    x_d, w_d = _nhwc(x).to(dev), _nhwc(w).to(dev)
    assert ops._small_conv_ok(3, 3, 1, 1, 1, C, Co, Nb * H * W)
    out = ops.conv2d(x_d, w_d, pad=1, shift=bias.to(dev), res=_nhwc(res).to(dev))
    _close(out, _nhwc((y + bias.view(1, -1, 1, 1) + res.float()).detach()), C * 9, "small conv fwd")
    generic = ops.conv2d(x_d, w_d, pad=1, shift=bias.to(dev), res=_nhwc(res).to(dev), tile=65)       # the tiled kernel on the same call
    assert float((out.float() - generic.float()).abs().max()) <= 2.0 ** -6 * float(generic.float().abs().max())
    dy = torch.randn(Nb, Co, H, W, generator=g).to(BF)
    y.backward(dy.float())
    if C in (16, 32) and Co in (8, 16):
        dw = ops.conv2d_wgrad(_nhwc(dy).to(dev), x_d, (Co, 3, 3, C), pad=1)
        wr2 = w.float().requires_grad_(True)
        F.conv2d(x.float(), wr2, padding=1).backward(dy.float())
        _close(dw, _nhwc(wr2.grad), Nb * H * W, "small conv wgrad", rtol=3e-3, atol_unit=2e-4)
    if Co in (8, 16, 32):
        prev = torch.randn(Nb, C, H, W, generator=g).to(BF)
        dx = ops.conv2d_dgrad(_nhwc(dy).to(dev), w_d, (H, W), pad=1, res=_nhwc(prev).to(dev))
        _close(dx, _nhwc(xr.grad + prev.float()), Co * 9, "small conv dgrad")


def test_attention_products(dev):
    from toist_amd import kernels as k, ops
    g = torch.Generator().manual_seed(5)
    B, H, Sq, Sk, dh = 2, 8, 100, 52, 32
    d = H * dh
    q, kk, v = _rand((B * Sq, d), g), _rand((B * Sk, d), g), _rand((B * Sk, d), g)
    scale = 1.0 / math.sqrt(dh)
    qf = q.float().view(B, Sq, H, dh).permute(0, 2, 1, 3)
    kf = kk.float().view(B, Sk, H, dh).permute(0, 2, 1, 3)
    vf = v.float().view(B, Sk, H, dh).permute(0, 2, 1, 3)
    s_ref = (qf @ kf.transpose(-1, -2)) * scale
    qd, kd, vd = q.to(dev), kk.to(dev), v.to(dev)
    s = ops.attn_scores(qd, kd, B, H, Sq, Sk, dh, scale)
    ld = s.shape[-1]
    _close(s.view(B, H, Sq, ld)[..., :Sk], s_ref, dh, "scores")
    key_pad = torch.zeros(B, Sk, dtype=torch.uint8)
    key_pad[1, -7:] = 1
    p = torch.empty_like(s)
    k.softmax_fwd(s, key_pad.to(dev), B, H, Sq, Sk, ld, p)
    s_used = s.float().cpu().view(B, H, Sq, ld)[..., :Sk].masked_fill(key_pad.bool()[:, None, None, :], float("-inf"))
    p_ref = torch.softmax(s_used, -1)
    pc = p.float().cpu().view(B, H, Sq, ld)
    assert float(pc[..., Sk:].abs().max()) == 0.0 if ld > Sk else True
    _close(pc[..., :Sk], p_ref, 1, "softmax", rtol=1e-2, atol_unit=2e-3)
    ctx = torch.empty(B * Sq, d, dtype=BF, device=dev)
    ops.attn_context(p, vd, B, H, Sq, Sk, dh, ctx)
    p_b = pc[..., :Sk]
    ctx_ref = (p_b @ vf).permute(0, 2, 1, 3).reshape(B * Sq, d)
    _close(ctx, ctx_ref, Sk, "context")
    # backward
    dctx = _rand((B * Sq, d), g)
    dq, dk, dv = (torch.empty(B * Sq, d, dtype=BF, device=dev), torch.empty(B * Sk, d, dtype=BF, device=dev),
                  torch.empty(B * Sk, d, dtype=BF, device=dev))

    def sm_bwd(dp):
        ds = torch.empty_like(dp)
        k.softmax_bwd(p, dp, B * H * Sq, Sk, ld, ds)
        return ds

    ops.attn_backward(p, scale, qd, kd, vd, dctx.to(dev), B, H, Sq, Sk, dh, dq, dk, dv, sm_bwd)
    qa, ka, va = (qf.clone().requires_grad_(True), kf.clone().requires_grad_(True), vf.clone().requires_grad_(True))
    sa = ((qa @ ka.transpose(-1, -2)) * scale).masked_fill(key_pad.bool()[:, None, None, :], float("-inf"))
    oa = torch.softmax(sa, -1) @ va
    oa.backward(dctx.float().view(B, Sq, H, dh).permute(0, 2, 1, 3))
    back = lambda t, S: t.permute(0, 2, 1, 3).reshape(B * S, d)
    _close(dv, back(va.grad, Sk), Sq, "dV", rtol=3e-2, atol_unit=6e-3)
    _close(dq, back(qa.grad, Sq), Sk, "dQ", rtol=3e-2, atol_unit=6e-3)
    _close(dk, back(ka.grad, Sk), Sq, "dK", rtol=3e-2, atol_unit=6e-3)


def test_rows(dev):
    from toist_amd import kernels as k
    g = torch.Generator().manual_seed(3)
    for rows, D, eps in [(37, 256, 1e-5), (130, 768, 1e-12)]:
        x = _rand((rows, D), g, 2.0)
        gamma, beta = torch.rand(D, generator=g) + 0.5, torch.randn(D, generator=g)
        xr = x.float().requires_grad_(True)
        gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
        y_ref = F.layer_norm(xr, (D,), gr, br, eps)
        xd = x.to(dev)
        y = torch.empty_like(xd)
        mean, rstd = torch.empty(rows, device=dev), torch.empty(rows, device=dev)
        k.layernorm_fwd(xd, gamma.to(dev), beta.to(dev), eps, y, mean, rstd)
        _close(y, y_ref.detach(), 1, "ln fwd", rtol=1e-2, atol_unit=1e-2)
        dy = _rand((rows, D), g)
        y_ref.backward(dy.float())
        dx = torch.empty_like(xd)
        dg, db = torch.zeros(D, device=dev), torch.zeros(D, device=dev)
        k.layernorm_bwd(dy.to(dev), xd, mean, rstd, gamma.to(dev), dx, dg, db)
        _close(dx, xr.grad, 1, "ln dx", rtol=2e-2, atol_unit=2e-2)
        _close(dg, gr.grad, rows, "ln dgamma", rtol=1e-2, atol_unit=1e-2)
        _close(db, br.grad, rows, "ln dbeta", rtol=1e-2, atol_unit=1e-2)
        # fused branch gradient: the same pass also emits dropout(dx) with the (seed, flat index) mask of the standalone kernel
        dx2, dxd = torch.empty_like(xd), torch.empty_like(xd)
        k.layernorm_bwd(dy.to(dev), xd, mean, rstd, gamma.to(dev), dx2, None, None, dx_drop=dxd, drop_p=0.25, seed=99)
        ref = torch.empty_like(xd)
        k.dropout(dx2, 0.25, 99, ref)
        assert torch.equal(dx2, dx)
        assert torch.equal(dxd == 0, ref == 0) and 0.2 < float((dxd == 0).float().mean()) < 0.3
        assert torch.allclose(dxd.float(), ref.float(), rtol=1e-2, atol=1e-6)        # masked from the unrounded value: <= 1 bf16 ulp apart
    a, b = _rand((64, 256), g), _rand((8, 256), g)
    out = torch.empty(64, 256, dtype=BF, device=dev)
    k.add(a.to(dev), b.to(dev), out, b_period=8 * 256)
    _close(out, a.float() + b.float().repeat(8, 1), 1, "add", rtol=1e-2, atol_unit=1e-2)
    # dropout: deterministic in (seed, index), keep-rate ~ 1-p, kept values scaled
    x = torch.ones(1 << 16, dtype=BF, device=dev)
    o1, o2 = torch.empty_like(x), torch.empty_like(x)
    k.dropout(x, 0.1, 1234, o1)
    k.dropout(x, 0.1, 1234, o2)
    assert torch.equal(o1, o2)
    keep = (o1 != 0).float().mean().item()
    assert abs(keep - 0.9) < 0.01, keep
    assert abs(float(o1.float().max()) - 1.0 / 0.9) < 1e-2


@pytest.mark.parametrize("N", [8, 16, 32, 128])
def test_colsum_tall_narrow(dev, N):
    """Bias gradient of a few-channel convolution over millions of pixels: the narrow column-sum kernel (all threads busy)."""
    from toist_amd import ops
    g = torch.Generator().manual_seed(N)
    x = torch.randn(70000, N, generator=g).to(BF)
    out = ops.bias_grad(x.to(dev))
    _close(out, x.float().sum(0), 70000, f"colsum N={N}", rtol=2e-3, atol_unit=2e-4)


@pytest.mark.parametrize("min_tiles", [1, 1 << 20])      # unsplit group / group that is also split along K
@pytest.mark.parametrize("R,stride", [(1, 1), (3, 1)])
def test_grouped_conv_wgrad_matches_separate_launches(dev, R, stride, min_tiles):
    """toist_group: the weight gradients of several same-shape convolutions as ONE unsplit launch (operands at arbitrary
    addresses, outputs / row scales at element offsets) against one launch per problem (split along K + fold) and fp32 math."""
    from toist_amd import ops
    g = torch.Generator().manual_seed(R)
    n, Nb, H, W, C, Co = 5, 2, 20, 24, 64, 128
    pad = 1 if R == 3 else 0
    flat = torch.zeros(n * Co * R * R * C + 64, dtype=torch.float32, device=dev)          # one gradient buffer, like ParamSet.flat
    scales = (torch.rand(n, Co, generator=g) + 0.5).to(dev)
    items, refs = [], []
    for i in range(n):
        x = torch.randn(Nb, H, W, C, generator=g).to(BF).to(dev)
        dy = torch.randn(Nb, H, W, Co, generator=g).to(BF).to(dev)
        out = flat[16 + i * Co * R * R * C: 16 + (i + 1) * Co * R * R * C].view(Co, R, R, C)
        items.append((dy, x, out, scales[i]))
        xr = x.float().permute(0, 3, 1, 2).cpu()
        w = torch.zeros(Co, C, R, R, requires_grad=True)
        torch.nn.functional.conv2d(xr, w, padding=pad).backward(dy.float().permute(0, 3, 1, 2).cpu())
        refs.append(w.grad.permute(0, 2, 3, 1) * scales[i].cpu()[:, None, None, None])
    old = ops.GROUP_MIN_TILES
    ops.GROUP_MIN_TILES = min_tiles
    try:
        ops.conv2d_wgrad_group(items, (Co, R, R, C), stride=stride, pad=pad)
        from toist_amd import kernels as k
        k.flush_reductions()
    finally:
        ops.GROUP_MIN_TILES = old
    grouped = [it[2].clone() for it in items]
    flat.zero_()
    for dy, x, out, rs in items:
        ops.conv2d_wgrad(dy, x, (Co, R, R, C), stride=stride, pad=pad, out=out, rscale=rs)
    assert float(flat[:16].abs().sum()) == 0 and float(flat[-48:].abs().sum()) == 0            # nothing written outside the slices
    for i in range(n):
        _close(grouped[i], refs[i], Nb * H * W, f"grouped wgrad {i}", rtol=1e-2, atol_unit=2e-3)
        assert torch.allclose(grouped[i], items[i][2], rtol=2e-3, atol=2e-2)


@pytest.mark.parametrize("B,H,S,dh", [(8, 12, 16, 64), (3, 4, 11, 32), (2, 2, 40, 64), (1, 3, 64, 16)])
def test_small_attention_against_autograd(dev, B, H, S, dh):
    """csrc/attn_small.hip (whole-head attention of the text encoder: S <= 64, head dim <= 64) forward and backward against fp32
    autograd, with key padding and the three projection biases added on load; dropout off (the mask is a hash, covered by the model tests)."""
    from toist_amd import kernels as k
    g = torch.Generator().manual_seed(S * dh)
    d = H * dh
    qkv = torch.randn(B * S, 3 * d, generator=g).to(BF)
    bias = [torch.randn(d, generator=g) * 0.3 for _ in range(3)]
    pad = torch.zeros(B, S, dtype=torch.uint8)
    for b in range(B):
        pad[b, S - (b % 3):] = 1 if b % 3 else 0
    dctx = torch.randn(B * S, d, generator=g).to(BF)
    scale = dh ** -0.5
    leaves = [(qkv[:, i * d:(i + 1) * d].float() + bias[i]).view(B, S, H, dh).transpose(1, 2).requires_grad_(True) for i in range(3)]
    sc = (leaves[0] @ leaves[1].transpose(-1, -2)) * scale
    sc = sc.masked_fill(pad.bool()[:, None, None, :], float("-inf"))
    ctx_ref = (sc.softmax(-1) @ leaves[2]).transpose(1, 2).reshape(B * S, d)
    ctx_ref.backward(dctx.float())
    ref_grads = [t.grad.transpose(1, 2).reshape(B * S, d) for t in leaves]
    qkv_d = qkv.to(dev)
    q, kk, v = (qkv_d[:, i * d:(i + 1) * d] for i in range(3))
    bd = [t.to(dev) for t in bias]
    ctx = torch.empty(B * S, d, dtype=BF, device=dev)
    stats = torch.empty(B * H * S * 2, dtype=torch.float32, device=dev)
    k.attn_small_fwd(q, kk, v, pad.to(dev), B, H, S, dh, scale, 0.0, 0, ctx, stats, *bd)
    _close(ctx, ctx_ref.detach(), S, "small attention forward", rtol=2e-2, atol_unit=3e-3)
    dqkv = torch.empty(B * S, 3 * d, dtype=BF, device=dev)
    dq, dk, dv = (dqkv[:, i * d:(i + 1) * d] for i in range(3))
    k.attn_small_bwd(q, kk, v, pad.to(dev), B, H, S, dh, scale, 0.0, 0, stats, dctx.to(dev), dq, dk, dv, *bd)
    for name, got, ref in zip("qkv", (dq, dk, dv), ref_grads):
        _close(got, ref, S * 4, f"small attention d{name}", rtol=3e-2, atol_unit=3e-3)


@pytest.mark.parametrize("M,N,K,drop_where,act", [(3328, 2048, 256, 2, 1), (3328, 256, 256, 1, 0), (800, 512, 128, 1, 1), (12800, 1024, 256, 0, 1), (1000, 264, 72, 2, 0)])
def test_short_k_panel_kernel_matches_generic_tiles(dev, M, N, K, drop_where, act):
    """csrc/gemm.hip panel_kernel (tile code 135: K <= 256, weight panel resident in registers, lean epilogue incl. the transformer's
    dropout) must reproduce the generic 64x64 tiles bit for bit -- same MFMA order along K, same epilogue arithmetic, same dropout hash."""
    from toist_amd import kernels as k, ops
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g).to(BF).to(dev)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(BF).to(dev)
    bias, res = torch.randn(N, generator=g).to(dev), torch.randn(M, N, generator=g).to(BF).to(dev)
    k.SEED_DEV = torch.full((1,), 12345, dtype=torch.int64, device=dev)
    kw = dict(res=res, act=k.ACT_RELU if act else k.ACT_NONE, drop_where=drop_where, drop_p=0.1 if drop_where else 0.0, drop_seed=99)
    a = ops.linear(x, w, bias, tile=65, **kw)
    b = ops.linear(x, w, bias, tile=135, **kw)
    assert torch.equal(a, b)
    wt = w.t().contiguous()
    aux = torch.randn(M, N, generator=g).to(BF).to(dev)
    da = ops.linear_dgrad(x, wt, res=res, act=k.ACT_MASK_POS, aux=aux, split_k=1)
    out = torch.empty_like(da)
    k.gemm(M, N, K, k.A_ROWK, k.operand(x, K), k.B_KROW, k.operand(wt, N), out, N, res=res, ldr=N, act=k.ACT_MASK_POS, aux=aux, ldaux=N, tile=135)
    assert torch.equal(da, out)


@pytest.mark.parametrize("tile", [65, 134, 130, 132, 129])
@pytest.mark.parametrize("M,N,K", [(1000, 256, 192), (333, 136, 1024)])
def test_lean_epilogue_equals_the_general_one(dev, tile, M, N, K):
    """epilogue_lean (bf16 output, picked by lean_epilogue_ok) against the general epilogue_tile, which the same GEMM takes when it
    stores f32: identical arithmetic, so rounding the f32 result to bf16 must give the lean result bit for bit -- bias, dropout
    (before the residual and after the activation), residual, ReLU / GELU / mask, on every tile with a lean instantiation (129 has none)."""
    from toist_amd import kernels as k, ops
    g = torch.Generator().manual_seed(tile + M)
    x = torch.randn(M, K, generator=g).to(BF).to(dev)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(BF).to(dev)
    bias, res = torch.randn(N, generator=g).to(dev), torch.randn(M, N, generator=g).to(BF).to(dev)
    aux = torch.randn(M, N, generator=g).to(BF).to(dev)
    k.SEED_DEV = torch.full((1,), 777, dtype=torch.int64, device=dev)
    for act, dw in ((k.ACT_RELU, 1), (k.ACT_GELU, 2), (k.ACT_NONE, 0)):
        kw = dict(res=res, act=act, drop_where=dw, drop_p=0.1 if dw else 0.0, drop_seed=5, tile=tile, split_k=1)
        lean = ops.linear(x, w, bias, **kw)
        full = ops.linear(x, w, bias, out_dtype=torch.float32, **kw)
        assert torch.equal(lean, full.to(BF)), f"act {act} dropout {dw}"
    wt = w.t().contiguous()
    ldn = wt.stride(0)
    out_b = torch.empty(M, N, dtype=BF, device=dev)
    out_f = torch.empty(M, N, dtype=torch.float32, device=dev)
    for out in (out_b, out_f):
        k.gemm(M, N, K, k.A_ROWK, k.operand(x, K), k.B_KROW, k.operand(wt, ldn), out, N, res=res, ldr=N, act=k.ACT_MASK_POS, aux=aux, ldaux=N, tile=tile)
    assert torch.equal(out_b, out_f.to(BF))


def _ulp_close(got, ref32, what):
    """bf16 result of an f32 accumulation in a different order: half a bf16 ulp of the f32 reference (2^-8 relative at the bottom of
    a binade) plus the order noise."""
    err = (got.float() - ref32).abs()
    bound = 4.0e-3 * ref32.abs() + 2e-4
    assert bool((err <= bound).all()), f"{what}: max excess {float((err - bound).max()):.3e}"


@pytest.mark.parametrize("M,N,K", [(12800, 256, 1024), (1000, 264, 320), (4096, 1024, 1088)])
def test_gemm128_kernel_plain(dev, M, N, K):
    """csrc/gemm.hip gemm128_kernel (tile code 136: 128 x 128 tiles, 64 x 64 wave tiles, the k-tile's halves on wave pairs, folded in
    the epilogue) against the generic tiles' f32 output: forward (row-major B, scale/shift/residual/ReLU) and data gradient (k-major
    B, residual, aux mask), ragged M/N tiles and k-tile counts that are not multiples of the 4-slot ring."""
    from toist_amd import kernels as k, ops
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g).to(BF).to(dev)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(BF).to(dev)
    bias, res = torch.randn(N, generator=g).to(dev), torch.randn(M, N, generator=g).to(BF).to(dev)
    aux = torch.randn(M, N, generator=g).to(BF).to(dev)
    for kw in (dict(res=res, act=k.ACT_RELU), dict(act=k.ACT_NONE)):
        got = ops.linear(x, w, bias, tile=136, split_k=1, **kw)
        ref = ops.linear(x, w, bias, tile=65, split_k=1, out_dtype=torch.float32, **kw)
        _ulp_close(got, ref, f"forward {kw['act']}")
    wt = w.t().contiguous()
    out_b = torch.empty(M, N, dtype=BF, device=dev)
    out_f = torch.empty(M, N, dtype=torch.float32, device=dev)
    for out, tile in ((out_b, 136), (out_f, 65)):
        k.gemm(M, N, K, k.A_ROWK, k.operand(x, K), k.B_KROW, k.operand(wt, N), out, N, res=res, ldr=N, act=k.ACT_MASK_POS, aux=aux, ldaux=N, tile=tile)
    _ulp_close(out_b, out_f, "dgrad")
    # directly against fp32 torch math on the same bf16-rounded operands (not through another HIP kernel)
    y32 = x.float().cpu() @ w.float().cpu().t() + bias.cpu()
    _ulp_close(ops.linear(x, w, bias, tile=136, split_k=1, res=res, act=k.ACT_RELU).cpu(), torch.relu(y32 + res.float().cpu()), "forward vs fp32 torch")
    _ulp_close(ops.linear(x, w, bias, tile=136, split_k=1).cpu(), y32, "plain forward vs fp32 torch")
    d32 = (x.float().cpu() @ wt.float().cpu() + res.float().cpu()) * (aux.float().cpu() > 0)
    _ulp_close(out_b.cpu(), d32, "dgrad vs fp32 torch")


@pytest.mark.parametrize("Nb,H,W,C,Co,R,pad,dil", [(8, 40, 40, 256, 256, 3, 1, 1), (2, 37, 43, 128, 256, 3, 2, 2), (3, 19, 23, 64, 192, 3, 1, 1), (2, 30, 30, 64, 128, 1, 0, 1)])
def test_gemm128_kernel_convolution_gathers(dev, Nb, H, W, C, Co, R, pad, dil):
    """gemm128_kernel as the stride-1 convolution gather (forward, FrozenBN scale/shift + ReLU) and as the transposed gather of the
    data gradient (reversed tap walk against the two-level k-major weights, aux mask): border taps, dilation, a ragged last row tile,
    18 / 9 k-tiles (not multiples of the ring), against the 64 x 64 tiles storing f32.  The first shape is layer 3 of the bench batch,
    where the dispatcher picks this kernel by itself: the tile-0 call must equal the explicit tile-136 call bit for bit."""
    from toist_amd import kernels as k, ops
    g = torch.Generator().manual_seed(Nb * H + C)
    x = torch.randn(Nb, H, W, C, generator=g).to(BF).to(dev)
    w = (torch.randn(Co, R, R, C, generator=g) / math.sqrt(R * R * C)).to(BF).to(dev)
    dy = torch.randn(Nb, H, W, Co, generator=g).to(BF).to(dev)
    aux = torch.randn(Nb, H, W, C, generator=g).to(BF).to(dev)
    res = torch.randn(Nb, H, W, Co, generator=g).to(BF).to(dev)
    scale, shift = (torch.rand(Co, generator=g) + 0.5).to(dev), torch.randn(Co, generator=g).to(dev)
    if R > 1:
        got = ops.conv2d(x, w, pad=pad, dil=dil, scale=scale, shift=shift, res=res, act=k.ACT_RELU, tile=136)
        ref = ops.conv2d(x, w, pad=pad, dil=dil, scale=scale, shift=shift, res=res, act=k.ACT_RELU, tile=65, out_dtype=torch.float32)
        _ulp_close(got, ref, "forward gather")
        dx = torch.empty(Nb, H, W, C, dtype=BF, device=dev)
        dx32 = torch.empty(Nb, H, W, C, dtype=torch.float32, device=dev)
        ops.conv2d_dgrad(dy, w, (H, W), pad=pad, dil=dil, act=k.ACT_MASK_POS, aux=aux, out=dx, tile=136)
        ops.conv2d_dgrad(dy, w, (H, W), pad=pad, dil=dil, act=k.ACT_MASK_POS, aux=aux, out=dx32, tile=65)
        _ulp_close(dx, dx32, "transposed gather")
        # stride 2 (conv2 of the first bottleneck of layers 2-4): forward gather only
        res2 = res[:, ::2, ::2].contiguous()
        got2 = ops.conv2d(x, w, stride=2, pad=pad, dil=dil, scale=scale, shift=shift, res=res2, act=k.ACT_RELU, tile=136)
        ref2 = ops.conv2d(x, w, stride=2, pad=pad, dil=dil, scale=scale, shift=shift, res=res2, act=k.ACT_RELU, tile=65, out_dtype=torch.float32)
        _ulp_close(got2, ref2, "strided forward gather")
        # directly against fp32 F.conv2d / autograd on the same bf16-rounded operands
        xc, wc = x.float().cpu().permute(0, 3, 1, 2), w.float().cpu().permute(0, 3, 1, 2)      # CPU fp32: no library convolution on the device
        aff = lambda t: t.permute(0, 2, 3, 1) * scale.cpu() + shift.cpu()
        _ulp_close(got.cpu(), torch.relu(aff(F.conv2d(xc, wc, padding=pad, dilation=dil)) + res.float().cpu()), "forward gather vs F.conv2d")
        _ulp_close(got2.cpu(), torch.relu(aff(F.conv2d(xc, wc, stride=2, padding=pad, dilation=dil)) + res2.float().cpu()), "strided forward gather vs F.conv2d")
        d32 = torch.nn.grad.conv2d_input(xc.shape, wc, dy.float().cpu().permute(0, 3, 1, 2), padding=pad, dilation=dil).permute(0, 2, 3, 1)
        _ulp_close(dx.cpu(), d32 * (aux.float().cpu() > 0), "transposed gather vs conv2d_input")
        if Nb * H * W >= 12800:
            assert torch.equal(ops.conv2d(x, w, pad=pad, dil=dil, scale=scale, shift=shift, res=res, act=k.ACT_RELU), got)
            assert torch.equal(ops.conv2d_dgrad(dy, w, (H, W), pad=pad, dil=dil, act=k.ACT_MASK_POS, aux=aux), dx)
    else:   # a 1x1 is a plain GEMM to ops.conv2d: K = 64 is below the kernel's minimum, the explicit tile must be refused loudly
        with pytest.raises(RuntimeError, match="128x128"):
            ops.conv2d(x, w, scale=scale, shift=shift, tile=136)


@pytest.mark.parametrize("Nb,H,W,C,Co,R,dil,n,min_tiles,expect", [
    (8, 16, 16, 1024, 256, 1, 1, 9, 1, 137),         # grouped 1x1, pairs pinned to XCDs (9 problems, the last XCD slots empty)
    (8, 16, 16, 128, 256, 3, 1, 9, 1, 137),          # grouped 3x3 gather: border taps, image boundaries inside k-tiles (256 pixels per image)
    (4, 24, 16, 128, 128, 3, 2, 3, 1 << 20, 137),    # grouped AND split along K (arena partials + batched fold), dilation 2
    (8, 40, 40, 256, 256, 1, 1, 1, 1, 0),            # one problem split along K by the host's heuristic (deferred fold): 137 or 138, whichever pays
    (2, 20, 12, 128, 128, 3, 1, 2, 1, -1),           # 480 pixels: K % 64 != 0 -> the generic tiles must take it (same call, same answer)
    (8, 16, 16, 1024, 256, 1, 1, 16, 1, 138),        # 256 x 128 block tiles: 16 grouped 1x1 problems of 1 x 8 tiles
    (4, 20, 16, 128, 512, 3, 1, 8, 1, 138),          # ... and the 3x3 gather with 32-pixel k-tiles (rows of 16 pixels, images of 320)
    (8, 24, 24, 256, 256, 3, 2, 2, 1 << 20, 138),    # ... grouped and split, dilation 2
])
def test_gemm128w_weight_gradient_kernel(dev, Nb, H, W, C, Co, R, dil, n, min_tiles, expect):
    """csrc/gemm.hip gemm128w_kernel (tile code 137: both operands k-major through ds_read_b64_tr_b16, table-free 3x3 gather, f32
    alpha * rscale * acc += into the gradient slice or k-slice partials, (problem, slice) pairs pinned to XCDs) as the dispatcher picks
    it for single and grouped weight gradients, and gemm256w_kernel (tile 138: the same with 256 x 128 block tiles, 32-pixel k-tiles, no
    k-fold), against the 64 x 64 tiles on the same calls: same products in f32, another summation order.  `expect` = the tile
    toist_gemm_pick_tile must report for the call (0: either of the two, -1: neither)."""
    from toist_amd import kernels as k, ops
    g = torch.Generator().manual_seed(Nb * H + C + R)
    pad = dil * (R // 2)
    flat = torch.zeros(n * Co * R * R * C + 64, dtype=torch.float32, device=dev)
    items = []
    for i in range(n):
        dy = torch.randn(Nb, H, W, Co, generator=g).to(BF).to(dev)
        x = torch.randn(Nb, H, W, C, generator=g).to(BF).to(dev)
        out = flat[16 + i * Co * R * R * C: 16 + (i + 1) * Co * R * R * C].view(Co, R, R, C)
        items.append((dy, x, out, (torch.rand(Co, generator=g) + 0.5).to(dev)))

    def run(tile):
        flat.copy_(base)
        old = (ops.GROUP_TILE, ops.GROUP_MIN_TILES, k.FORCE_TILE)
        ops.GROUP_TILE, ops.GROUP_MIN_TILES, k.FORCE_TILE = tile, min_tiles, tile
        try:
            if n == 1:
                dy, x, out, rs = items[0]
                ops.conv2d_wgrad(dy, x, out.shape, pad=pad, dil=dil, out=out, rscale=rs, defer=True)
            else:
                ops.conv2d_wgrad_group(items, items[0][2].shape, pad=pad, dil=dil)
            k.flush_reductions()
        finally:
            ops.GROUP_TILE, ops.GROUP_MIN_TILES, k.FORCE_TILE = old
        return flat.clone()

    base = torch.randn(flat.shape, generator=g).to(dev) * 0.1          # accumulate = True: the kernel adds to what is there
    ref = run(65)
    bk = k.B_CONVX if R == 3 else k.B_KROW
    k.PROFILE = {"key": frozenset({(137, k.A_KROW, bk), (138, k.A_KROW, bk)}), "records": [], "other": {}}
    try:
        got = run(0)
        picked = [r[3][0] for r in k.PROFILE["records"]]
    finally:
        k.PROFILE = None
    assert torch.equal(got[:16], base[:16]) and torch.equal(got[-48:], base[-48:])               # nothing written outside the slices
    assert float((got - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
    # directly against fp32 torch (conv2d_weight on the same bf16-rounded operands), problem by problem
    for i, (dy, x, out, rs) in enumerate(items[:3]):
        w32 = torch.nn.grad.conv2d_weight(x.float().cpu().permute(0, 3, 1, 2), (Co, C, R, R), dy.float().cpu().permute(0, 3, 1, 2), padding=pad, dilation=dil)
        want = w32.permute(0, 2, 3, 1) * rs.cpu()[:, None, None, None] + base[16 + i * Co * R * R * C: 16 + (i + 1) * Co * R * R * C].view(Co, R, R, C).cpu()
        mine = got[16 + i * Co * R * R * C: 16 + (i + 1) * Co * R * R * C].view(Co, R, R, C).cpu()
        assert float((mine - want).abs().max()) <= 2e-3 * float(want.abs().max()), f"problem {i} vs fp32 conv2d_weight"
    if expect == -1:
        assert picked == [] and torch.equal(got, ref)
    elif expect == 0:
        assert len(picked) == 1
    else:
        assert picked == [expect], f"the dispatcher picked {picked}"


@pytest.mark.parametrize("N,H,W", [(2, 128, 160), (1, 71, 93), (3, 40, 24)])
def test_fused_stem_matches_the_three_launch_path(dev, N, H, W):
    """csrc/stem.hip (conv 7x7/2 + FrozenBN shift + ReLU + max-pool 3x3/2 from the fp32 NCHW image in one launch) against pack_image +
    the implicit-GEMM convolution + maxpool3x3s2 on the same bf16 weights: same products, f32 sums in another order, so the pooled bf16
    values agree to a bf16 ulp; and against fp32 torch on the bf16-rounded operands.  Odd sizes: ragged tiles, image borders inside tiles."""
    from toist_amd import kernels as k, ops
    g = torch.Generator().manual_seed(H + W)
    img = torch.randn(N, 3, H, W, generator=g).to(dev)
    w = (torch.randn(64, 7, 7, 3, generator=g) / math.sqrt(147)).to(dev)
    w8 = torch.nn.functional.pad(w, (0, 5)).to(BF).contiguous()
    shift = torch.randn(64, generator=g).to(dev) * 0.3
    CH, CW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    PH, PW = (CH - 1) // 2 + 1, (CW - 1) // 2 + 1
    out = torch.empty(N, PH, PW, 64, dtype=BF, device=dev)
    k.stem_fwd(img, w8, shift, out)
    xin = torch.empty(N, H, W, 8, dtype=BF, device=dev)
    k.pack_image(img, xin)
    y = ops.conv2d(xin, w8, stride=2, pad=3, shift=shift, act=k.ACT_RELU, cin_real=3)
    ref = torch.empty(N, PH, PW, 64, dtype=BF, device=dev)
    k.maxpool3x3s2(y, ref)
    err = (out.float() - ref.float()).abs()
    assert bool((err <= 8e-3 * ref.float().abs() + 1e-3).all()), float(err.max())
    t = torch.nn.functional.conv2d(img.to(BF).float(), w8[..., :3].float().permute(0, 3, 1, 2), stride=2, padding=3) + shift.view(1, -1, 1, 1)
    t = torch.nn.functional.max_pool2d(torch.relu(t), 3, 2, 1).permute(0, 2, 3, 1)
    err = (out.float() - t).abs()
    assert bool((err <= 8e-3 * t.abs() + 2e-3).all()), float(err.max())


@pytest.mark.parametrize("Nb,H,W,C,Co,R,split", [(8, 32, 32, 128, 256, 3, 8), (8, 64, 64, 256, 512, 1, 16), (4, 48, 80, 128, 128, 3, 15)])
def test_gemm128w_strided_weight_gradients(dev, Nb, H, W, C, Co, R, split):
    """Stride-2 weight gradients (conv2 and the downsample convolution of the first bottleneck of a ResNet stage) on gemm128w_kernel /
    gemm256w_kernel: the source pixel of an output pixel is rebuilt per k-tile from the lane's running (image, y, x); against the 64 x 64
    tiles on the same call (split along K by the host's rule, deferred fold) and against fp32 autograd."""
    from toist_amd import kernels as k, ops
    g = torch.Generator().manual_seed(H + C + R)
    pad = R // 2
    OH, OW = (H + 2 * pad - R) // 2 + 1, (W + 2 * pad - R) // 2 + 1
    x = torch.randn(Nb, H, W, C, generator=g).to(BF).to(dev)
    dy = torch.randn(Nb, OH, OW, Co, generator=g).to(BF).to(dev)
    rs = (torch.rand(Co, generator=g) + 0.5).to(dev)

    def run(tile):
        out = torch.zeros(Co, R, R, C, dtype=torch.float32, device=dev)
        old = k.FORCE_TILE
        k.FORCE_TILE = tile
        try:
            ops.conv2d_wgrad(dy, x, (Co, R, R, C), stride=2, pad=pad, out=out, rscale=rs, defer=True, split_k=split)   # slices as the step's tape picks them
            k.flush_reductions()
        finally:
            k.FORCE_TILE = old
        return out

    ref = run(65)
    k.PROFILE = {"key": frozenset({(137, k.A_KROW, k.B_CONVX), (138, k.A_KROW, k.B_CONVX)}), "records": [], "other": {}}
    try:
        got = run(0)
        picked = [r[3][0] for r in k.PROFILE["records"]]
    finally:
        k.PROFILE = None
    assert len(picked) == 1, "the dispatcher did not pick the 128-wide weight-gradient kernels"
    assert float((got - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
    w = torch.zeros(Co, C, R, R, requires_grad=True)
    torch.nn.functional.conv2d(x.float().permute(0, 3, 1, 2).cpu(), w, stride=2, padding=pad).backward(dy.float().permute(0, 3, 1, 2).cpu())
    want = w.grad.permute(0, 2, 3, 1) * rs.cpu()[:, None, None, None]
    assert float((got.cpu() - want).abs().max()) <= 2e-3 * float(want.abs().max())
