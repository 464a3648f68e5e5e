"""GPU parity of the segmentation branch (config 3) against the oracle (pinned to the reference by
tests/golden/segm.npz): attention map + mask head forward (rel err <= 4e-2, bf16 vs fp32), parameter and
input gradients (cosine >= 0.97), and the fused mask losses (focal + dice) forward / backward (rtol 1e-3)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-12))


def cos(a, b):
    return float(torch.nn.functional.cosine_similarity(a.float().cpu().flatten(), b.float().cpu().flatten(), dim=0))


def test_mask_head_forward_backward(dev):
    from oracle import model_ref
    from toist_amd.segmentation import DETRsegm, MaskHeadSmallConv, MHAttentionMap
    torch.manual_seed(0)
    B, Q, d, H, h, w = 2, 6, 256, 8, 5, 6

    class Stub(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.transformer = type("T", (), {"d_model": d, "nhead": H})()
    seg = DETRsegm(Stub(), "smallconv", freeze_detr=False)
    g = torch.Generator().manual_seed(1)
    for n, p in seg.named_parameters():
        if n.endswith("bias") or "gn" in n:
            p.data.add_(torch.randn(p.shape, generator=g) * 0.05)
    sd = {k_: v.detach().clone().float().requires_grad_(True) for k_, v in seg.state_dict().items()}
    seg.to(dev)
    hs = (torch.randn(B, Q, d, generator=g)).to(BF)
    mem = (torch.randn(B, h * w, d, generator=g)).to(BF)
    src = (torch.randn(B, h * w, d, generator=g)).to(BF)
    c4 = torch.randn(B, 2 * h, 2 * w, 1024, generator=g).clamp(min=0).to(BF)
    c3 = torch.randn(B, 4 * h, 4 * w, 512, generator=g).clamp(min=0).to(BF)
    c2 = torch.randn(B, 8 * h, 8 * w, 256, generator=g).clamp(min=0).to(BF)
    fmask = torch.zeros(B, h, w, dtype=torch.bool)
    fmask[1, :, 4:] = True
    ins = [t.to(dev).requires_grad_(True) for t in (hs.view(B * Q, d), mem.view(B * h * w, d), src.view(B * h * w, d), c4, c3, c2)]
    masks = seg._masks(*ins, fmask.to(dev), B, Q, h, w)
    gout = torch.randn(masks.shape, generator=g) * 0.1
    masks.backward(gout.to(dev))
    torch.cuda.synchronize()
    # oracle (fp32 on the same bf16-rounded inputs)
    r = [t.float().requires_grad_(True) for t in (hs, mem, src, c4, c3, c2)]
    nchw = lambda t: t.permute(0, 3, 1, 2)
    bm = model_ref.attention_map(sd, "bbox_attention.", r[0], r[1].transpose(1, 2).reshape(B, d, h, w), fmask, H)
    ref = model_ref.mask_head(sd, "mask_head.", r[2].transpose(1, 2).reshape(B, d, h, w), bm, [nchw(r[3]), nchw(r[4]), nchw(r[5])])
    ref = ref.view(B, Q, 8 * h, 8 * w)
    ref.backward(gout)
    assert rel(masks, ref) < 4e-2, rel(masks, ref)
    bad = {}
    top = max(float(v.grad.norm()) for v in sd.values())
    for n, p in seg.named_parameters():
        rn = float(sd[n].grad.norm())
        if rn < 1e-5 * top:
            # analytically zero gradient (k_linear.bias: a constant key shift leaves every softmax unchanged):
            # only bf16 round-off may remain
            assert float(p.grad.float().norm()) < 1e-2 * top, n
            continue
        c, ratio = cos(p.grad, sd[n].grad), float(p.grad.float().norm().cpu() / (rn + 1e-20))
        if c < 0.97 or not (0.8 < ratio < 1.25):
            bad[n] = (round(c, 4), round(ratio, 3))
    assert not bad, bad
    names = ["hs", "memory", "src_proj", "c4", "c3", "c2"]
    for n, a, b in zip(names, ins, r):
        got = a.grad.float().cpu().reshape(b.grad.shape)
        assert cos(got, b.grad) > 0.97, (n, cos(got, b.grad))


def test_mask_losses(dev):
    from oracle import model_ref
    from toist_amd.matcher import MatchResult
    from toist_amd.segmentation import mask_losses
    g = torch.Generator().manual_seed(3)
    B, Q, hm, wm = 2, 7, 24, 32
    pred = (torch.randn(B, Q, hm, wm, generator=g) * 2).requires_grad_(True)
    sizes = [2, 3]
    targets = [{"masks": torch.rand(t, 96 - 8 * i, 128, generator=g) > 0.6, "boxes": torch.zeros(t, 4)} for i, t in enumerate(sizes)]
    indices = [(torch.tensor([1, 4]), torch.tensor([1, 0])), (torch.tensor([0, 2, 6]), torch.tensor([2, 0, 1]))]
    ref = model_ref.loss_masks(pred, targets, indices, 5.0)
    (ref["loss_mask"] * 1.5 + ref["loss_dice"] * 0.7).backward()
    src = torch.cat([i for i, _ in indices])[None].to(dev)
    tgt = torch.cat([j for _, j in indices])[None].to(dev)
    match = MatchResult(src, tgt, torch.zeros(B, dtype=torch.int32), sizes, Q)
    p2 = pred.detach().to(dev).requires_grad_(True)
    t_dev = [{k_: v.to(dev) for k_, v in t.items()} for t in targets]
    got = mask_losses({"pred_masks": p2}, t_dev, match, 0, torch.tensor(5.0, device=dev))
    (got["loss_mask"] * 1.5 + got["loss_dice"] * 0.7).backward()
    for k_ in ("loss_mask", "loss_dice"):
        assert abs(float(got[k_]) - float(ref[k_])) <= 1e-4 * abs(float(ref[k_])) + 1e-6, (k_, float(got[k_]), float(ref[k_]))
    err = float((p2.grad.cpu() - pred.grad).abs().max())
    assert err <= 1e-3 * float(pred.grad.abs().max()) + 1e-8, err


@pytest.mark.parametrize("VH,VW,TH,TW", [(120, 180, 128, 192), (134, 201, 192, 256), (96, 128, 96, 128)])
def test_mask_losses_of_a_batch_do_not_depend_on_its_bucket(dev, VH, VW, TH, TW):
    """harness.CapturedTrainStep pads a batch to its bucket's (Hp, Wp); the reference resizes the [ceil(H/4), ceil(W/4)]
    prediction to the batch's own largest image (H, W) and averages loss_mask over H * W (/root/reference/models/mdetr.py:839-851).  With
    StaticTargets.valid_hw = {VH, VW, hs, ws} the bucket-sized call must give EXACTLY what the batch-sized call gives on the corresponding
    corners -- the reference-pinned path of test_mask_losses -- values and gradients, with zero gradient outside the corner."""
    from oracle import model_ref
    from toist_amd.segmentation import _MaskLossFn
    g = torch.Generator().manual_seed(VH + TW)
    B, Q = 2, 5
    hs, ws, h, w = (VH + 3) // 4, (VW + 3) // 4, TH // 4, TW // 4
    pred_big = (torch.randn(B * Q, h, w, generator=g) * 2)
    sizes = [2, 1]
    gt_small = (torch.rand(sum(sizes), VH, VW, generator=g) > 0.6)
    gt_big = torch.zeros(sum(sizes), TH, TW, dtype=torch.bool)
    gt_big[:, :VH, :VW] = gt_small
    gt_big[:, VH:, :] = True                      # garbage outside the batch's corner must not matter
    gt_big[:, :, VW:] = True
    indices = [(torch.tensor([1, 4]), torch.tensor([1, 0])), (torch.tensor([3]), torch.tensor([0]))]
    # the reference-pinned oracle on the batch-sized tensors
    p_ref = pred_big[:, :hs, :ws].reshape(B, Q, hs, ws).clone().requires_grad_(True)
    targets = [{"masks": gt_small[:2], "boxes": torch.zeros(2, 4)}, {"masks": gt_small[2:], "boxes": torch.zeros(1, 4)}]
    ref = model_ref.loss_masks(p_ref, targets, indices, 3.0)
    (ref["loss_mask"] * 1.5 + ref["loss_dice"] * 0.7).backward()
    pred_row = torch.tensor([1, 4, Q + 3, -1], dtype=torch.int32, device=dev)          # one unused slot, as in a fixed-capacity pair table
    gt_row = torch.tensor([1, 0, 2, 0], dtype=torch.int32, device=dev)
    p2 = pred_big.to(dev).requires_grad_(True)
    valid = torch.tensor([VH, VW, hs, ws], dtype=torch.int32, device=dev)
    vals = _MaskLossFn.apply(p2, pred_row, gt_big.to(torch.uint8).to(dev), gt_row, torch.tensor(3.0, device=dev), TH, TW, None, None, valid)
    (vals[0] * 1.5 + vals[1] * 0.7).backward()
    assert abs(float(vals[0]) - float(ref["loss_mask"])) <= 1e-4 * abs(float(ref["loss_mask"])) + 1e-6, (float(vals[0]), float(ref["loss_mask"]))
    assert abs(float(vals[1]) - float(ref["loss_dice"])) <= 1e-4 * abs(float(ref["loss_dice"])) + 1e-6
    got = p2.grad.cpu().reshape(B, Q, h, w)
    assert float((got[:, :, :hs, :ws] - p_ref.grad).abs().max()) <= 1e-3 * float(p_ref.grad.abs().max()) + 1e-8
    outside = got.clone()
    outside[:, :, :hs, :ws] = 0
    assert float(outside.abs().max()) == 0.0


@pytest.mark.parametrize("N,HW,C", [(6, 400, 264), (4, 1600, 64), (3, 6400, 16)])
def test_groupnorm_backward_rederives_the_relu_mask(dev, N, HW, C):
    """toist_groupnorm_bwd without y (round 5: the mask y > 0 of relu(GroupNorm(x)) re-derived from x, stats, gamma, beta -- two passes over
    the activation fewer) against the same call with y, and against fp32 autograd of F.relu(F.group_norm(x, 8)) (segmentation.py:203-241)."""
    from toist_amd import kernels as k
    BF = torch.bfloat16
    g = torch.Generator().manual_seed(N * HW + C)
    x = torch.randn(N, HW, C, generator=g).to(BF).to(dev)
    dy = torch.randn(N, HW, C, generator=g).to(BF).to(dev)
    gamma = (1.0 + 0.2 * torch.randn(C, generator=g)).to(dev)
    beta = (0.3 * torch.randn(C, generator=g)).to(dev)
    y = torch.empty_like(x)
    stats = torch.zeros(N, 8, 2, device=dev)
    k.groupnorm_fwd(x, gamma, beta, N, HW, C, 8, 1e-5, True, y, stats)
    outs = []
    for with_y in (True, False):
        dx = torch.empty_like(x)
        dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
        bstats = torch.empty(N, 8, 2, device=dev)
        k.groupnorm_bwd(dy, y if with_y else None, x, stats, gamma, N, HW, C, 8, 1e-5, True, dx, dg, db, bstats, beta=None if with_y else beta)
        outs.append((dx, dg, db))
    # the two masks differ at most where the normalised value is a rounding error away from zero (a flipped element also moves its group's sums by one term)
    for a, b in zip(outs[0], outs[1]):
        assert (a.float() - b.float()).norm() / a.float().norm() < 2e-3
    xr = x.float().permute(0, 2, 1).reshape(N, C, HW, 1).requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    ref = torch.relu(torch.nn.functional.group_norm(xr, 8, gr, br, 1e-5))
    ref.backward(dy.float().permute(0, 2, 1).reshape(N, C, HW, 1))
    dx_ref = xr.grad.reshape(N, C, HW).permute(0, 2, 1)
    dx = outs[1][0].float()
    assert (dx - dx_ref).norm() / dx_ref.norm() < 1e-2
    assert (outs[1][1] - gr.grad).norm() / gr.grad.norm() < 1e-2 and (outs[1][2] - br.grad).norm() / br.grad.norm() < 1e-2


@pytest.mark.parametrize("extra_consumer", [False, "after", "before"])
def test_matched_only_backward_equals_dense(dev, extra_consumer):
    """The mask losses read pred_masks[src_idx] only (/root/reference/models/mdetr.py:827-853), so the mask head's backward runs on the
    matched maps alone; every parameter / input gradient must equal the dense backward's (which multiplies the zeros through).  With a
    second consumer of pred_masks the program has to notice and fall back to the dense path -- WHICHEVER consumer was created first
    ("before": the extra consumer's gradient reaches autograd's input buffer after the mask losses' and is accumulated onto it; round 5's
    address-compared dense sentinel was added into in place there and the extra gradient was silently dropped)."""
    from toist_amd import segmentation
    from toist_amd.matcher import MatchResult
    from toist_amd.segmentation import DETRsegm, mask_losses
    B, Q, d, H, h, w = 3, 7, 256, 8, 5, 6

    class Stub(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.transformer = type("T", (), {"d_model": d, "nhead": H})()
    torch.manual_seed(0)
    seg = DETRsegm(Stub(), "smallconv", freeze_detr=False).to(dev)
    g = torch.Generator().manual_seed(5)
    base = [torch.randn(B * Q, d, generator=g), torch.randn(B * h * w, d, generator=g), torch.randn(B * h * w, d, generator=g),
            torch.randn(B, 2 * h, 2 * w, 1024, generator=g).clamp(min=0), torch.randn(B, 4 * h, 4 * w, 512, generator=g).clamp(min=0),
            torch.randn(B, 8 * h, 8 * w, 256, generator=g).clamp(min=0)]
    fmask = torch.zeros(B, h, w, dtype=torch.bool, device=dev)
    sizes = [2, 0, 3]                                    # an image without targets in the middle
    targets = [{"masks": (torch.rand(t, 160, 192, generator=g) > 0.6).to(dev), "boxes": torch.zeros(t, 4, device=dev)} for t in sizes]
    src = torch.tensor([[5, 1, 0, 2, 6]], device=dev)    # image 0: queries 5, 1; image 2: queries 0, 2, 6
    tgt = torch.tensor([[1, 0, 2, 0, 1]], device=dev)
    match = MatchResult(src, tgt, torch.zeros(B, dtype=torch.int32), sizes, Q)

    ran = []
    from toist_amd import kernels as _k
    orig_rows = _k.upsample_add_rows

    def spy(*a, **kw):
        ran.append(1)
        return orig_rows(*a, **kw)

    def run(flag):
        segmentation.MATCHED_ONLY_BACKWARD = flag
        _k.upsample_add_rows = spy
        try:
            seg.zero_grad(set_to_none=True)
            ins = [t.to(BF).to(dev).requires_grad_(True) for t in base]
            masks = seg._masks(*ins, fmask, B, Q, h, w)
            assert hasattr(masks, "toist_matched_rows") == flag
            extra = (masks * masks).mean() * 0.05 if extra_consumer == "before" else None
            out = mask_losses({"pred_masks": masks}, targets, match, 0, torch.tensor(5.0, device=dev))
            loss = out["loss_mask"] * 1.5 + out["loss_dice"] * 0.7
            if extra_consumer == "after":
                extra = (masks * masks).mean() * 0.05
            if extra is not None:
                loss = loss + extra
            ran.clear()
            loss.backward()
            assert (len(ran) > 0) == (flag and not extra_consumer), (flag, extra_consumer, ran)   # the matched-rows path ran iff nobody else consumed pred_masks
            torch.cuda.synchronize()
            return {n: p.grad.detach().float().clone() for n, p in seg.named_parameters()}, [t.grad.detach().float().clone() for t in ins]
        finally:
            segmentation.MATCHED_ONLY_BACKWARD = True
            _k.upsample_add_rows = orig_rows
    pd, idn = run(False)
    ps, isp = run(True)
    top = max(float(v.norm()) for v in pd.values())
    for n in pd:
        err = float((pd[n] - ps[n]).norm())
        assert err <= 2e-2 * float(pd[n].norm()) + 2e-5 * top, (n, err, float(pd[n].norm()))
    # the loss gradient is summed with f32 atomics (run-to-run order), so bf16 roundings downstream may flip: direction and size, not bits
    for n, a, b in zip(["hs", "memory", "src_proj", "c4", "c3", "c2"], idn, isp):
        ratio = float(b.norm() / a.norm())
        assert cos(a, b) > 0.9995 and 0.99 < ratio < 1.01, (n, cos(a, b), ratio)


def test_fused_tail_equals_per_op_launches(dev):
    """lay4 / lay5 / out_lay as one launch each (csrc/maskstage.hip: GroupNorm + ReLU and the 2x upsample + FPN term applied on the way into the
    3x3, statistics in the epilogue) against the per-op launches: same roundings, so logits agree to the bf16 rounding the per-op path puts
    on out_lay's output, and every gradient agrees in direction and size.  Odd tile counts (40 x 48 and 20 x 24 outputs: partial 16 x 16 tiles)."""
    from toist_amd import segmentation
    from toist_amd.segmentation import DETRsegm
    B, Q, d, H, h, w = 2, 5, 256, 8, 5, 6

    class Stub(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.transformer = type("T", (), {"d_model": d, "nhead": H})()
    torch.manual_seed(0)
    seg = DETRsegm(Stub(), "smallconv", freeze_detr=False)
    g = torch.Generator().manual_seed(11)
    for n, p in seg.named_parameters():
        if n.endswith("bias") or "gn" in n:
            p.data.add_(torch.randn(p.shape, generator=g) * 0.05)
    seg.to(dev)
    base = [torch.randn(B * Q, d, generator=g), torch.randn(B * h * w, d, generator=g), torch.randn(B * h * w, d, generator=g),
            torch.randn(B, 2 * h, 2 * w, 1024, generator=g).clamp(min=0), torch.randn(B, 4 * h, 4 * w, 512, generator=g).clamp(min=0),
            torch.randn(B, 8 * h, 8 * w, 256, generator=g).clamp(min=0)]
    fmask = torch.zeros(B, h, w, dtype=torch.bool, device=dev)
    gout = (torch.randn(B, Q, 8 * h, 8 * w, generator=g) * 0.1).to(dev)

    def run(flag):
        segmentation.FUSED_TAIL = flag
        try:
            seg.zero_grad(set_to_none=True)
            ins = [t.to(BF).to(dev).requires_grad_(True) for t in base]
            masks = seg._masks(*ins, fmask, B, Q, h, w)
            masks.backward(gout)
            torch.cuda.synchronize()
            return masks.detach().float(), {n: p.grad.detach().float().clone() for n, p in seg.named_parameters()}, [t.grad.detach().float().clone() for t in ins]
        finally:
            segmentation.FUSED_TAIL = True
    m0, p0, i0 = run(False)
    m1, p1, i1 = run(True)
    assert float((m0 - m1).abs().max()) <= 2 ** -7 * float(m0.abs().max()), float((m0 - m1).abs().max())
    assert rel(m1, m0) < 1e-2, rel(m1, m0)      # the fused stages take the FPN term out of the convolution (linearity): one rounding less, placed elsewhere
    top = max(float(v.norm()) for v in p0.values())
    for n in p0:
        if float(p0[n].norm()) < 1e-5 * top:
            continue
        ratio = float(p1[n].norm() / p0[n].norm())
        assert cos(p0[n], p1[n]) > 0.995 and 0.95 < ratio < 1.05, (n, cos(p0[n], p1[n]), ratio)     # two bf16 paths with their roundings in different places (bias gradients are sums with heavy cancellation)
    for n, a, b in zip(["hs", "memory", "src_proj", "c4", "c3", "c2"], i0, i1):
        ratio = float(b.norm() / a.norm())
        assert cos(a, b) > 0.995 and 0.95 < ratio < 1.05, (n, cos(a, b), ratio)


@pytest.mark.parametrize("stage", ["lay4", "lay5", "out_lay"])
def test_mask_stage_kernel_against_torch(dev, stage):
    """toist_mask_stage_fwd against the same operators in fp32 torch (segmentation.py:223-240 of the reference: GroupNorm(8, C) + ReLU ->
    nearest 2x -> + FPN term -> 3x3 conv; the FPN term enters as fpn_conv = lay(adapter(fpn)) by linearity), ragged sizes (partial 16 x 16 tiles),
    several images: output within bf16 rounding of the fp32 result, GroupNorm sums of the output within 1e-3."""
    import torch.nn.functional as F
    from toist_amd import kernels as k
    g = torch.Generator().manual_seed({"lay4": 1, "lay5": 2, "out_lay": 3}[stage])
    B, Q = 2, 3
    N = B * Q
    cin, cout, gn_in, up = {"lay4": (64, 32, False, True), "lay5": (32, 16, True, True), "out_lay": (16, 1, True, False)}[stage]
    H, W = (40, 56) if up else (24, 40)
    SH, SW = (H // 2, W // 2) if up else (H, W)
    src = (torch.randn(N, SH, SW, cin, generator=g) * 1.5 + 0.3).to(BF)
    if not gn_in:
        src = src.clamp(min=0)
    w = (torch.randn(cout, 3, 3, cin, generator=g) * (2.0 / (9 * cin)) ** 0.5).to(BF)
    bias = torch.randn(cout, generator=g) * 0.1
    gamma, beta = 1 + 0.2 * torch.randn(cin, generator=g), 0.2 * torch.randn(cin, generator=g)
    fpn_conv = (torch.randn(B, H, W, cout, generator=g)).to(BF) if up else None
    # reference, fp32 on the same bf16 inputs
    x = src.float().permute(0, 3, 1, 2)
    if gn_in:
        x = F.relu(F.group_norm(x, 8, gamma, beta, 1e-5))
    if up:
        x = F.interpolate(x, scale_factor=2, mode="nearest")
    ref = F.conv2d(x, w.float().permute(0, 3, 1, 2), None if up else bias, padding=1)
    if up:
        ref = ref + fpn_conv.float().permute(0, 3, 1, 2).repeat_interleave(Q, 0)
    ref = ref.permute(0, 2, 3, 1)                                   # [N,H,W,cout]
    # kernel
    d = lambda t: None if t is None else t.to(dev)
    st_in = None
    if gn_in:
        st_in = torch.empty(N, 8, 2, dtype=torch.float32, device=dev)
        k.groupnorm_fwd(d(src), d(gamma), d(beta), N, SH * SW, cin, 8, 1e-5, True, None, st_in)      # statistics only
    if cout == 1:
        out = torch.empty(N, H, W, dtype=torch.float32, device=dev)
        k.mask_stage_fwd(d(src), st_in, d(gamma), d(beta), None, d(w), d(bias), out, None, N, Q, H, W, cin, 1, 1, True, False)
        torch.cuda.synchronize()
        err = float((out.cpu() - ref[..., 0]).abs().max())
        assert err <= 2e-2 * float(ref.abs().max()), err
        assert rel(out, ref[..., 0]) < 1e-2
        return
    out = torch.empty(N, H, W, cout, dtype=BF, device=dev)
    st = torch.empty(N, 8, 2, dtype=torch.float32, device=dev)
    k.mask_stage_fwd(d(src), st_in, d(gamma) if gn_in else None, d(beta) if gn_in else None, d(fpn_conv), d(w), None, out, st, N, Q, H, W, cin, cout, cout, gn_in, up)
    torch.cuda.synchronize()
    assert rel(out, ref) < 1e-2, rel(out, ref)
    o = out.float().cpu().view(N, H * W, 8, cout // 8)
    want = torch.stack([o.sum((1, 3)), (o * o).sum((1, 3))], -1)    # sums of the ROUNDED output, as the per-op statistics pass would see it
    assert torch.allclose(st.cpu(), want, rtol=1e-3, atol=1e-2), float((st.cpu() - want).abs().max())


def test_segment_sum_and_row_variants(dev):
    """toist_sum_segments / toist_upsample_add_rows / toist_groupnorm_apply (the matched-rows backward's helpers) against torch."""
    import torch.nn.functional as F
    from toist_amd import kernels as k
    g = torch.Generator().manual_seed(7)
    B, Q, n, Hs, Ws, C = 3, 5, 6, 6, 10, 32
    x = torch.randn(n, Hs, Ws, C, generator=g).to(BF)
    seg = torch.tensor([0, 2, 2, 6], dtype=torch.int32)
    out = torch.empty(B, Hs, Ws, C, dtype=BF, device=dev)
    k.sum_segments(x.to(dev), seg.to(dev), B, n, Hs * Ws * C, out)
    want = torch.stack([x[0:2].float().sum(0), torch.zeros(Hs, Ws, C), x[2:6].float().sum(0)])
    assert torch.allclose(out.float().cpu(), want, atol=2e-2, rtol=1e-2)
    rows = torch.tensor([1, 4, 5, 7, 12, 14], dtype=torch.int64)         # maps of images 0, 0, 1, 1, 2, 2
    fpn = torch.randn(B, 2 * Hs, 2 * Ws, C, generator=g).to(BF)
    up = torch.empty(n, 2 * Hs, 2 * Ws, C, dtype=BF, device=dev)
    k.upsample_add_rows(x.to(dev), fpn.to(dev), rows.to(dev), n, Q, Hs, Ws, C, up)
    want = (x.float().repeat_interleave(2, 1).repeat_interleave(2, 2) + fpn.float()[rows // Q]).to(BF)
    assert torch.equal(up.cpu(), want)
    gamma, beta = 1 + 0.2 * torch.randn(C, generator=g), 0.2 * torch.randn(C, generator=g)
    st = torch.empty(n, 8, 2, dtype=torch.float32, device=dev)
    y0 = torch.empty(n, Hs, Ws, C, dtype=BF, device=dev)
    k.groupnorm_fwd(x.to(dev), gamma.to(dev), beta.to(dev), n, Hs * Ws, C, 8, 1e-5, True, y0, st)
    y1 = torch.empty_like(y0)
    k.groupnorm_apply(x.to(dev), st, gamma.to(dev), beta.to(dev), n, Hs * Ws, C, 8, 1e-5, True, y1)
    torch.cuda.synchronize()
    assert torch.equal(y0, y1)
    ref = F.relu(F.group_norm(x.float().permute(0, 3, 1, 2), 8, gamma, beta, 1e-5)).permute(0, 2, 3, 1)
    assert rel(y1, ref) < 1e-2


@pytest.mark.parametrize("sizes", [[2, 0, 3], [0, 0, 0]])
def test_static_pair_table_matches_the_eager_pairs(dev, sizes):
    """mask_losses_static (fixed-capacity pair table, live slots computed on the device: the hipGraph replay's path) against mask_losses on the same
    assignment: same losses, same gradients through the matched-rows backward -- including a batch WITHOUT any target (every slot unused: zero losses,
    zero gradients, nothing non-finite)."""
    from types import SimpleNamespace
    from toist_amd.matcher import MatchResult
    from toist_amd.segmentation import DETRsegm, mask_losses, mask_losses_static
    B, Q, d, H, h, w = 3, 7, 256, 8, 5, 6
    TH, TW = 160, 192

    class Stub(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.transformer = type("T", (), {"d_model": d, "nhead": H})()
    torch.manual_seed(0)
    seg = DETRsegm(Stub(), "smallconv", freeze_detr=False).to(dev)
    g = torch.Generator().manual_seed(9)
    base = [torch.randn(B * Q, d, generator=g), torch.randn(B * h * w, d, generator=g), torch.randn(B * h * w, d, generator=g),
            torch.randn(B, 2 * h, 2 * w, 1024, generator=g).clamp(min=0), torch.randn(B, 4 * h, 4 * w, 512, generator=g).clamp(min=0),
            torch.randn(B, 8 * h, 8 * w, 256, generator=g).clamp(min=0)]
    fmask = torch.zeros(B, h, w, dtype=torch.bool, device=dev)
    M = sum(sizes)
    cap_per = 4
    gt_list = [(torch.rand(t, TH, TW, generator=g) > 0.6) for t in sizes]
    src_all = [torch.tensor([5, 1]), torch.tensor([], dtype=torch.long), torch.tensor([0, 2, 6])]
    tgt_all = [torch.tensor([1, 0]), torch.tensor([], dtype=torch.long), torch.tensor([2, 0, 1])]
    src = torch.cat([s[:t] for s, t in zip(src_all, sizes)])[None].to(dev)
    tgt = torch.cat([s[:t] for s, t in zip(tgt_all, sizes)])[None].to(dev)
    match = MatchResult(src, tgt, torch.zeros(B, dtype=torch.int32), sizes, Q)
    # the static image of the same batch: masks packed at stride cap_per per image, offsets on the device
    st_masks = torch.zeros(B * cap_per, TH, TW, dtype=torch.uint8, device=dev)
    for i, m in enumerate(gt_list):
        st_masks[i * cap_per:i * cap_per + m.shape[0]] = m.to(torch.uint8).to(dev)
    off = [0]
    for t in sizes:
        off.append(off[-1] + t)
    st = SimpleNamespace(cap=B * cap_per, masks=st_masks, mask_hw=(TH, TW), match_off=torch.tensor(off, dtype=torch.int32, device=dev),
                         tgt_off=torch.tensor([i * cap_per for i in range(B)], dtype=torch.int32, device=dev), num_boxes=torch.tensor([5.0], device=dev))
    pad = torch.zeros(1, st.cap - M, dtype=torch.long, device=dev)
    match_st = SimpleNamespace(src=torch.cat([src, pad], 1), tgt=torch.cat([tgt, pad], 1))

    def run(static):
        seg.zero_grad(set_to_none=True)
        ins = [t.to(BF).to(dev).requires_grad_(True) for t in base]
        masks = seg._masks(*ins, fmask, B, Q, h, w)
        if static:
            out = mask_losses_static({"pred_masks": masks}, st, match_st, 0, 1)
        else:
            targets = [{"masks": m.to(dev), "boxes": torch.zeros(m.shape[0], 4, device=dev)} for m in gt_list]
            out = mask_losses({"pred_masks": masks}, targets, match, 0, torch.tensor(5.0, device=dev))
        (out["loss_mask"] * 1.5 + out["loss_dice"] * 0.7).backward()
        torch.cuda.synchronize()
        return {k_: float(v) for k_, v in out.items()}, {n: p.grad.detach().float().clone() for n, p in seg.named_parameters()}, [t.grad.detach().float().clone() for t in ins]
    ls, ps, gs = run(True)
    for v in list(ps.values()) + gs:
        assert torch.isfinite(v).all()
    if M == 0:
        assert ls["loss_mask"] == 0.0 and ls["loss_dice"] == 0.0
        assert all(float(v.abs().max()) == 0.0 for v in ps.values()) and all(float(v.abs().max()) == 0.0 for v in gs)
        return
    le, pe, ge = run(False)
    for k_ in ("loss_mask", "loss_dice"):
        assert abs(ls[k_] - le[k_]) <= 1e-4 * abs(le[k_]) + 1e-6, (k_, ls[k_], le[k_])
    top = max(float(v.norm()) for v in pe.values())
    for n in pe:
        err = float((pe[n] - ps[n]).norm())
        assert err <= 2e-2 * float(pe[n].norm()) + 2e-5 * top, (n, err, float(pe[n].norm()))
    for n, a, b in zip(["hs", "memory", "src_proj", "c4", "c3", "c2"], ge, gs):
        ratio = float(b.norm() / a.norm())
        assert cos(a, b) > 0.9995 and 0.99 < ratio < 1.01, (n, cos(a, b), ratio)


@pytest.mark.parametrize("H,W,OH,OW", [(4, 6, 8, 12), (8, 12, 15, 23), (15, 23, 30, 45), (7, 5, 13, 10), (25, 42, 50, 84), (50, 84, 100, 167)])
def test_resize_add_matches_torch_nearest(dev, H, W, OH, OW):
    """toist_resize_add / toist_resize_add_bwd (round 6: FPN levels of any size) against `fpn + F.interpolate(x, size=..., mode="nearest")` and its autograd
    (/root/reference/models/segmentation.py:218, 225, 232) -- exact: both only move and add bf16 values (the sums of <= 9 terms are formed in f32 here)."""
    from toist_amd import kernels as k
    g = torch.Generator().manual_seed(H * 131 + OW)
    B, Q, C = 2, 3, 16
    x = torch.randn(B * Q, H, W, C, generator=g).to(BF)
    f = torch.randn(B, OH, OW, C, generator=g).to(BF)
    out = torch.empty(B * Q, OH, OW, C, dtype=BF, device=dev)
    k.resize_add(x.to(dev), f.to(dev), None, B * Q, Q, H, W, OH, OW, C, out)
    xr = x.float().permute(0, 3, 1, 2).requires_grad_(True)
    up = torch.nn.functional.interpolate(xr, size=(OH, OW), mode="nearest")
    ref = (up + f.float().permute(0, 3, 1, 2).repeat_interleave(Q, 0)).permute(0, 2, 3, 1)
    assert torch.equal(out.float().cpu(), ref.detach().to(BF).float())
    rows = torch.tensor([4, 1, 5], dtype=torch.int64, device=dev)              # a gathered subset: maps 4, 1, 5 (images 1, 0, 1)
    sub = torch.empty(3, OH, OW, C, dtype=BF, device=dev)
    k.resize_add(x.to(dev)[rows].contiguous(), f.to(dev), rows, 3, Q, H, W, OH, OW, C, sub)
    assert torch.equal(sub, out[rows])
    dy = torch.randn(B * Q, OH, OW, C, generator=g).to(BF)
    dx = torch.empty(B * Q, H, W, C, dtype=BF, device=dev)
    k.resize_add_bwd(dy.to(dev), B * Q, H, W, OH, OW, C, dx)
    up.backward(dy.float().permute(0, 3, 1, 2))
    assert torch.equal(dx.float().cpu(), xr.grad.permute(0, 2, 3, 1).to(BF).float())
