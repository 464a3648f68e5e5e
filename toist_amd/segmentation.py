"""Segmentation branch of TOIST (config 3) on the MI355X kernels.

Mirrors /root/reference/models/segmentation.py: DETRsegm (:17-168), MaskHeadSmallConv (:170-241),
MHAttentionMap (:244-273), dice_loss / sigmoid_focal_loss (:276-319) and SetCriterion.loss_masks
(/root/reference/models/mdetr.py:827-853) -- same module tree / state_dict names (`detr.*`,
`bbox_attention.{q,k}_linear.*`, `mask_head.lay1..5/gn1..5/out_lay/adapter1..3.*`).

MI355X-first decisions: feature maps stay NHWC bf16; the 256 `src_proj` channels of the mask head's first
convolution are identical for the 100 queries of an image, so that part is convolved once per image and
broadcast through the GEMM epilogue (only the 8 attention-map channels are convolved per query: 33x
fewer MACs in lay1); FPN adapters are computed per image and added inside the 2x-upsample kernel; the
backward of every per-query broadcast is a sum over queries followed by per-image GEMMs.
"""
from collections import OrderedDict

import torch
from torch import nn

from . import engine, functions
from . import kernels as k
from . import ops
from .backbone import nearest_mask
from .misc import NestedTensor

BF16 = torch.bfloat16


class MHAttentionMap(nn.Module):
    """Parameter holder of the 2-D attention map (segmentation.py:244-273)."""

    def __init__(self, query_dim, hidden_dim, num_heads, dropout=0, bias=True):
        super().__init__()
        self.num_heads, self.hidden_dim = num_heads, hidden_dim
        self.q_linear = nn.Linear(query_dim, hidden_dim, bias=bias)
        self.k_linear = nn.Linear(query_dim, hidden_dim, bias=bias)
        nn.init.zeros_(self.k_linear.bias)
        nn.init.zeros_(self.q_linear.bias)
        nn.init.xavier_uniform_(self.k_linear.weight)
        nn.init.xavier_uniform_(self.q_linear.weight)
        self.normalize_fact = float(hidden_dim / num_heads) ** -0.5


class MaskHeadSmallConv(nn.Module):
    """Parameter holder of the FPN mask decoder (segmentation.py:170-241)."""

    def __init__(self, dim, fpn_dims, context_dim):
        super().__init__()
        inter = [dim, context_dim // 2, context_dim // 4, context_dim // 8, context_dim // 16, context_dim // 64]
        self.inter_dims = inter
        self.lay1, self.gn1 = nn.Conv2d(dim, dim, 3, padding=1), nn.GroupNorm(8, dim)
        self.lay2, self.gn2 = nn.Conv2d(dim, inter[1], 3, padding=1), nn.GroupNorm(8, inter[1])
        self.lay3, self.gn3 = nn.Conv2d(inter[1], inter[2], 3, padding=1), nn.GroupNorm(8, inter[2])
        self.lay4, self.gn4 = nn.Conv2d(inter[2], inter[3], 3, padding=1), nn.GroupNorm(8, inter[3])
        self.lay5, self.gn5 = nn.Conv2d(inter[3], inter[4], 3, padding=1), nn.GroupNorm(8, inter[4])
        self.out_lay = nn.Conv2d(inter[4], 1, 3, padding=1)
        self.dim = dim
        self.adapter1 = nn.Conv2d(fpn_dims[0], inter[1], 1)
        self.adapter2 = nn.Conv2d(fpn_dims[1], inter[2], 1)
        self.adapter3 = nn.Conv2d(fpn_dims[2], inter[3], 1)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_uniform_(m.weight, a=1)
                nn.init.constant_(m.bias, 0)


# The mask losses (mdetr.py:827-853) read `outputs["pred_masks"][src_idx]`: only the matched queries' maps carry a gradient, and every
# operator of the mask head acts map by map (per-map convolutions, GroupNorm(8, C) statistics per sample, per-map upsampling; the FPN /
# image terms are broadcast over the queries).  The gradient of every unmatched map is therefore exactly zero from out_lay back to lay1,
# and the backward program runs on the matched maps alone (T of B*Q = 800).  The hand-off below carries (gradient rows, row indices,
# per-image slot ranges) from _MaskLossFn.backward to the mask program's tape; any other consumer of pred_masks keeps the dense path.
MATCHED_ONLY_BACKWARD = True
# lay4 / lay5 / out_lay as one launch each (csrc/maskstage.hip) when the head has the reference's widths (64 -> 32 -> 16 -> 1); False = per-op launches
FUSED_TAIL = True


class _MatchedRows:
    """Hand-off between `_MaskLossFn.backward` and the mask program's backward: pair t's gradient in row t of `grad`, the prediction rows in
    `rows`, and the zero SENTINEL autograd carried in place of the dense gradient (held here, so autograd can never accumulate into it in
    place; it is a stride-0 expansion of one element, so no dense sum can alias it either)."""
    __slots__ = ("grad", "rows", "seg", "sentinel", "version")

    def __init__(self):
        self.clear()

    def clear(self):
        self.grad = self.rows = self.seg = self.sentinel = None
        self.version = -1


_ZERO = {}


def _zero_sentinel(dev):
    """One f32 zero per device.  `z.expand(shape)` is the gradient autograd carries when only the matched rows are non-zero: no kernel ever
    reads or writes it (the 82 MB `zeros_like(pred)` of round 5 is gone), and anything autograd adds to it lands in a NEW dense tensor."""
    z = _ZERO.get(str(dev))
    if z is None:
        z = _ZERO[str(dev)] = torch.zeros(1, dtype=torch.float32, device=dev)
    return z


def _krsc_bf16(w):
    """[O,I,R,S] fp32 -> contiguous [O,R,S,I] bf16 (the GEMM B operand of an NHWC convolution)."""
    return w.detach().permute(0, 2, 3, 1).contiguous().to(BF16)


def _add_conv_grad(param_grad, tmp_krsc, c_lo=0, c_hi=None):
    """param_grad [O,I,R,S] (fp32, any strides) += tmp [O,R,S,Cpart] for input channels [c_lo, c_hi)."""
    c_hi = param_grad.shape[1] if c_hi is None else c_hi
    param_grad[:, c_lo:c_hi].add_(tmp_krsc.permute(0, 3, 1, 2))


def _conv_gn_relu(tape, x, W, b, G_w, G_b, shape, *, extra_res=None, res_bcast=None, w_slice=None, pick=None):
    """y = relu(GN8(conv3x3(x) + bias [+ broadcast residual])).  x: Var NHWC; returns Var NHWC.
    W: ParamView of the [O,I,3,3] weight; w_slice = (lo, hi) restricts the input channels used."""
    N, H, Wd, C = shape
    wk = W.w if w_slice is None else W.w[..., w_slice[0]:w_slice[1]].contiguous()
    Co = wk.shape[0]
    pre = ops.conv2d(x.data, wk, pad=1, shift=(b.f32 if extra_res is None else None), res=extra_res, res_bcast=res_bcast)
    stats = torch.empty(N, 8, 2, dtype=torch.float32, device=pre.device)
    y = torch.empty_like(pre)
    k.groupnorm_fwd(pre, G_w.f32, G_b.f32, N, H * Wd, Co, 8, 1e-5, True, y, stats)
    out = engine.Var(y)
    pre_var = engine.Var(pre)  # gradient w.r.t. the conv output (after GN backward)

    def bwd():
        g = out.take_grad()
        if g is None:
            return
        dpre = torch.empty_like(g)
        n = g.shape[0]                       # N, or the number of matched maps (pick gathers their rows of the saved tensors)
        pre_s, stats_s, x_s = (pre, stats, x.data) if pick is None else (pick(pre), pick(stats), pick(x.data))
        bstats = torch.empty(n, 8, 2, dtype=torch.float32, device=g.device)
        k.groupnorm_bwd(g, None, pre_s, stats_s, G_w.f32, n, H * Wd, Co, 8, 1e-5, True, dpre, G_w.g, G_b.g if G_w.g is not None else None, bstats, beta=G_b.f32)
        pre_var.grad = dpre
        if W.g is not None:
            tmp = ops.conv2d_wgrad(dpre, x_s, wk.shape, pad=1)
            _add_conv_grad(W.g, tmp, *(w_slice or (0, None)))
            if extra_res is None and b.g is not None:
                ops.bias_grad(dpre.view(-1, Co), out=b.g)
        if x.needs_grad:
            x.grad = ops.conv2d_dgrad(dpre, wk, (H, Wd), pad=1, res=x.grad)

    tape.record(bwd)
    return out, pre_var


class DETRsegm(nn.Module):
    def __init__(self, detr, mask_head="smallconv", freeze_detr=False):
        super().__init__()
        self.detr = detr
        if freeze_detr:
            for p in self.parameters():
                p.requires_grad_(False)
        hidden_dim, nheads = detr.transformer.d_model, detr.transformer.nhead
        self.bbox_attention = MHAttentionMap(hidden_dim, hidden_dim, nheads, dropout=0)
        if mask_head != "smallconv":
            raise RuntimeError(f"Unknown mask model {mask_head}")
        self.mask_head = MaskHeadSmallConv(hidden_dim + nheads, [1024, 512, 256], hidden_dim)
        self._cache = {}

    # ---- the mask program ---------------------------------------------------------------------------------
    def _masks(self, hs_last, memory, src_proj, c4, c3, c2, feat_mask, B, Q, h, w):
        """hs_last [B*Q,d], memory / src_proj [B*h*w, d], c4/c3/c2 NHWC bf16 -> pred mask logits [B,Q,8h,8w] f32."""
        H = self.bbox_attention.num_heads
        d = self.bbox_attention.hidden_dim
        dh = d // H
        HW = h * w
        ld = ops.round8(HW)
        scale = self.bbox_attention.normalize_fact
        named = OrderedDict(("bbox_attention." + n, p) for n, p in self.bbox_attention.named_parameters())
        named.update(("mask_head." + n, p) for n, p in self.mask_head.named_parameters())
        transforms = {n: _krsc_bf16 for n, p in named.items() if p.dim() == 4}
        key_pad = feat_mask.flatten(1).to(torch.uint8).contiguous()
        dev = hs_last.device

        BQ = B * Q
        sel = {"rows": None, "scatter": None, "seg": None}     # set by out_bwd when only the matched maps carry a gradient

        def pick(t):
            """Rows of a saved [B*Q, ...] tensor that the backward needs (all of them on the dense path)."""
            return t if sel["rows"] is None else t.index_select(0, sel["rows"])

        def image_sum(g, per, out):
            """out[b] = sum of the maps of image b: all Q of them, or the matched ones (packed image by image)."""
            if sel["rows"] is None:
                k.sum_queries(g, B, Q, per, out)
            else:
                k.sum_segments(g, sel["seg"], B, g.shape[0], per, out)

        def prog(tape, ps, hs, mem, src, f4, f3, f2):
            A = lambda n: ps["bbox_attention." + n]
            M = lambda n: ps["mask_head." + n]
            # -- attention map (segmentation.py:262-273)
            q = engine.linear_chain(tape, hs, [(A("q_linear.weight"), A("q_linear.bias"), k.ACT_NONE, False)])
            kk = engine.linear_chain(tape, mem, [(A("k_linear.weight"), A("k_linear.bias"), k.ACT_NONE, False)])
            scores = torch.empty(B, Q, H, ld, dtype=BF16, device=dev)
            k.gemm(Q, HW, dh, k.A_ROWK, k.operand(q.data, d, bs_outer=Q * d, bs_inner=dh), k.B_ROWK, k.operand(kk.data, d, bs_outer=HW * d, bs_inner=dh),
                   scores, H * ld, batch=B * H, batch_inner=H, cs_outer=Q * H * ld, cs_inner=ld, alpha=scale, tile=64)
            prob = torch.empty(B * Q, h, w, H, dtype=BF16, device=dev)
            k.attnmap_softmax_fwd(scores, key_pad, B, Q, H, HW, ld, prob)
            del scores
            pv = engine.Var(prob)

            def att_bwd():
                g = pv.take_grad()
                if g is None:
                    return
                if sel["rows"] is not None:          # matched maps only: back to one row per (image, query); unused slots land in row B*Q
                    gd = torch.zeros(BQ + 1, h, w, H, dtype=BF16, device=dev)
                    gd.index_copy_(0, sel["scatter"], g)
                    g = gd[:BQ]
                ds = torch.empty(B, Q, H, ld, dtype=BF16, device=dev)
                k.attnmap_softmax_bwd(prob, g, B * Q, H, HW, ld, ds)
                dq = torch.empty(B * Q, d, dtype=BF16, device=dev)
                dk = torch.empty(B * HW, d, dtype=BF16, device=dev)
                # dq[b,q,n,:] = scale * sum_p ds[b,q,n,p] kk[b,p,n,:] ; dk[b,p,n,:] = scale * sum_q ds[b,q,n,p] q[b,q,n,:]
                k.gemm(Q, dh, HW, k.A_ROWK, k.operand(ds, H * ld, bs_outer=Q * H * ld, bs_inner=ld), k.B_KROW,
                       k.operand(kk.data, d, bs_outer=HW * d, bs_inner=dh), dq, d, batch=B * H, batch_inner=H, cs_outer=Q * d, cs_inner=dh,
                       alpha=scale, tile=64)
                k.gemm(HW, dh, Q, k.A_KROW, k.operand(ds, H * ld, bs_outer=Q * H * ld, bs_inner=ld), k.B_KROW,
                       k.operand(q.data, d, bs_outer=Q * d, bs_inner=dh), dk, d, batch=B * H, batch_inner=H, cs_outer=HW * d, cs_inner=dh,
                       alpha=scale, tile=64)
                q.grad, kk.grad = dq, dk

            tape.record(att_bwd)

            # -- lay1: image part once per image, attention channels per query, added in the GEMM epilogue
            W1, b1 = M("lay1.weight"), M("lay1.bias")
            src4 = engine.Var(src.data.view(B, h, w, d), needs_grad=src.needs_grad)
            w1_img = W1.w[..., :d].contiguous()
            y_img = ops.conv2d(src4.data, w1_img, pad=1, shift=b1.f32)            # [B,h,w,264]
            a1, pre1 = _conv_gn_relu(tape, pv, W1, b1, M("gn1.weight"), M("gn1.bias"), (B * Q, h, w, H), extra_res=y_img.view(B * HW, -1),
                                     res_bcast=(Q * HW, HW), w_slice=(d, d + H), pick=pick)

            C1 = w1_img.shape[0]

            def img_bwd():
                g = pre1.take_grad()
                if g is None:
                    return
                gsum = torch.empty(B, h, w, C1, dtype=BF16, device=dev)
                image_sum(g, HW * C1, gsum)
                if W1.g is not None:
                    tmp = ops.conv2d_wgrad(gsum, src4.data, w1_img.shape, pad=1)
                    _add_conv_grad(W1.g, tmp, 0, d)
                    ops.bias_grad(gsum.view(-1, C1), out=b1.g)
                if src4.needs_grad:
                    gs = ops.conv2d_dgrad(gsum, w1_img, (h, w), pad=1)
                    engine.accumulate(src, gs.view(B * HW, d))

            # the block's backward (recorded inside _conv_gn_relu) must run BEFORE img_bwd: insert img_bwd just before it
            tape.steps.insert(len(tape.steps) - 1, img_bwd)

            a2, _ = _conv_gn_relu(tape, a1, M("lay2.weight"), M("lay2.bias"), M("gn2.weight"), M("gn2.bias"), (B * Q, h, w, C1), pick=pick)

            def adapter(feat, i):
                """The FPN term of stage i (adapter{i}, a 1x1 convolution of the backbone feature): [B*2H*2W, Cx] bf16."""
                Wa, ba = M(f"adapter{i}.weight"), M(f"adapter{i}.bias")
                Cx, Cin_f = Wa.w.shape[0], feat.data.shape[-1]
                wa = Wa.w.view(Cx, Cin_f)
                return ops.linear(feat.data.view(-1, Cin_f), wa, ba.f32), wa

            def up_grads(g, x, feat, i, wa, H_, W_):
                """Backward of `adapter{i}(feat) + nearest_resize(x)` for the gradient g [n,OH,OW,Cx] of the sum (OH x OW = the FPN level's size)."""
                Wa, ba = M(f"adapter{i}.weight"), M(f"adapter{i}.bias")
                Cx, Cin_f = wa.shape
                n, OH, OW = g.shape[0], g.shape[1], g.shape[2]
                if x.needs_grad:
                    gx = torch.empty(n, H_, W_, Cx, dtype=BF16, device=dev)
                    k.resize_add_bwd(g, n, H_, W_, OH, OW, Cx, gx)
                    engine.accumulate(x, gx)
                if Wa.g is not None or feat.needs_grad:
                    gf = torch.empty(B * OH * OW, Cx, dtype=BF16, device=dev)
                    image_sum(g, OH * OW * Cx, gf)
                    if Wa.g is not None:
                        tmp = torch.zeros(Cx, Cin_f, dtype=torch.float32, device=dev)
                        ops.linear_wgrad(gf, feat.data.view(-1, Cin_f), out=tmp, bias_out=ba.g)
                        Wa.g.add_(tmp.view(Cx, Cin_f, 1, 1))
                    if feat.needs_grad:
                        gfeat = ops.linear_dgrad(gf, wa)
                        engine.accumulate(feat, gfeat.view(feat.data.shape))

            def fpn_stage(x, feat, i, H_, W_):
                """x [BQ,H_,W_,C] -> relu(GN(lay(adapter(feat) + nearest_resize(x)))) at the FPN level's own size (segmentation.py:216-234: 2H_ x 2W_
                for image sides that are multiples of 32, otherwise whatever the backbone's ceil-divisions left)"""
                Cx = x.data.shape[-1]
                OH, OW = feat.data.shape[1], feat.data.shape[2]
                f, wa = adapter(feat, i)                                             # [B*OH*OW, Cx]
                up = torch.empty(B * Q, OH, OW, Cx, dtype=BF16, device=dev)
                k.resize_add(x.data, f, None, B * Q, Q, H_, W_, OH, OW, Cx, up)
                uv = engine.Var(up)

                def up_bwd():
                    g = uv.take_grad()
                    if g is not None:
                        up_grads(g, x, feat, i, wa, H_, W_)

                tape.record(up_bwd)
                return _conv_gn_relu(tape, uv, M(f"lay{i + 2}.weight"), M(f"lay{i + 2}.bias"), M(f"gn{i + 2}.weight"), M(f"gn{i + 2}.bias"),
                                     (B * Q, OH, OW, Cx), pick=pick)[0]

            def matched_rows_of(g):
                """g = gradient of the mask logits [B,Q,8h,8w] exactly as autograd delivered it (Var.raw_grad) -> the rows the backward has to run on
                (sets `sel`).  "Nothing but the mask losses consumed pred_masks" is decided by IDENTITY of the zero sentinel `_MaskLossFn.backward`
                returned: same address, every stride zero, version counter untouched.  The sink holds a reference to the sentinel, so autograd's
                input buffer cannot add another consumer's gradient into it in place (it only does that to a tensor it owns alone), whichever
                consumer was created first: any sum is a new dense tensor and takes the dense path."""
                sel["rows"] = sel["scatter"] = sel["seg"] = None
                zero = sink.sentinel
                if sink.grad is not None:
                    rows = sink.rows
                    if (zero is not None and g.data_ptr() == zero.data_ptr() and not any(g.stride()) and zero._version == sink.version):
                        # the gradient is sink.grad in rows `rows`, zero elsewhere
                        sel["rows"] = rows.clamp(min=0).to(torch.int64)
                        sel["scatter"] = torch.where(rows < 0, torch.full_like(rows, BQ), rows).to(torch.int64)
                        sel["seg"] = sink.seg
                        g = sink.grad
                    else:                        # another consumer added its gradient: dense backward of the sum (unused slots hold zeros)
                        g = g.to(torch.float32).contiguous().view(BQ, Hm, Wm).index_add(0, rows.clamp(min=0).to(torch.int64), sink.grad)
                    sink.clear()
                    return g
                return g.to(torch.float32).contiguous().view(BQ, Hm, Wm)

            def out_lay_grads(g, a5_rows, Wo, bo, wo8):
                """Backward of out_lay (one output channel, padded to 8 for the GEMM operands) for the logit gradient g [n,Hm,Wm] f32."""
                n = g.shape[0]
                g8 = torch.zeros(n, Hm, Wm, 8, dtype=BF16, device=dev)
                g8[..., 0] = g.to(BF16)
                if Wo.g is not None:
                    tmp = ops.conv2d_wgrad(g8, a5_rows, wo8.shape, pad=1)
                    _add_conv_grad(Wo.g, tmp[:1])
                    bo.g.add_(g.sum().reshape(1))
                return g8

            # sizes of the three FPN levels the maps are resized to; the logits have the last one's (C2: ceil(side / 4))
            (h4, w4), (h3, w3), (Hm, Wm) = f4.data.shape[1:3], f3.data.shape[1:3], f2.data.shape[1:3]
            doubling = (h4, w4, h3, w3, Hm, Wm) == (2 * h, 2 * w, 4 * h, 4 * w, 8 * h, 8 * w)      # image sides are multiples of 32
            a3 = fpn_stage(a2, f4, 1, h, w)
            Wo, bo = M("out_lay.weight"), M("out_lay.bias")
            W4, W5 = M("lay4.weight"), M("lay5.weight")
            C3, C4, C5 = a3.data.shape[-1], W4.w.shape[0], W5.w.shape[0]
            wo8 = torch.zeros(8, 3, 3, C5, dtype=BF16, device=dev)   # out_lay: Cout = 1 padded to 8 output channels (16-byte rows for the backward GEMM operands)
            wo8[:1] = Wo.w
            if FUSED_TAIL and doubling and (C3, C4, C5) == (64, 32, 16) and W4.w.shape[-1] == 64 and W5.w.shape[-1] == 32 and Wo.w.shape[0] == 1:
                # ---- lay4 / lay5 / out_lay as one launch each (csrc/maskstage.hip): the upsampled sums, the normalised activations and the padded
                # out_lay output are never written; the backward re-creates them for the maps it runs on (the matched ones)
                b4, b5 = M("lay4.bias"), M("lay5.bias")
                g4w, g4b, g5w, g5b = M("gn4.weight"), M("gn4.bias"), M("gn5.weight"), M("gn5.bias")
                H4, W4_, H5, W5_ = 4 * h, 4 * w, 8 * h, 8 * w
                fa, wa2 = adapter(f3, 2)
                pre4 = torch.empty(BQ, H4, W4_, C4, dtype=BF16, device=dev)
                st4 = torch.empty(BQ, 8, 2, dtype=torch.float32, device=dev)
                # the convolution is linear: lay(adapter(fpn) + up2(x)) = lay(adapter(fpn)) [one small convolution per IMAGE, bias included] + lay_nobias(up2(x))
                fa_conv = ops.conv2d(fa.view(B, H4, W4_, C3), W4.w, pad=1, shift=b4.f32)
                k.mask_stage_fwd(a3.data, None, None, None, fa_conv, W4.w, None, pre4, st4, BQ, Q, H4, W4_, C3, C4, C4, False, True)
                fb, wa3 = adapter(f2, 3)
                pre5 = torch.empty(BQ, H5, W5_, C5, dtype=BF16, device=dev)
                st5 = torch.empty(BQ, 8, 2, dtype=torch.float32, device=dev)
                fb_conv = ops.conv2d(fb.view(B, H5, W5_, C4), W5.w, pad=1, shift=b5.f32, tile=64)   # 8 images: the tiled implicit GEMM (the direct few-channel kernel is sized for 800 maps: 71 us here)
                k.mask_stage_fwd(pre4, st4, g4w.f32, g4b.f32, fb_conv, W5.w, None, pre5, st5, BQ, Q, H5, W5_, C4, C5, C5, True, True)
                masks = torch.empty(B, Q, H5, W5_, dtype=torch.float32, device=dev)
                k.mask_stage_fwd(pre5, st5, g5w.f32, g5b.f32, None, Wo.w, bo.f32, masks, None, BQ, Q, H5, W5_, C5, 1, 1, True, False)
                mv = engine.Var(masks)
                mv.raw_grad = True

                def norm_rows(pre, st, gw, gb, HW_, C):
                    """relu(GroupNorm(pre)) of the rows the backward runs on, from the forward's statistics."""
                    x_s, st_s = pick(pre), pick(st)
                    y = torch.empty_like(x_s)
                    k.groupnorm_apply(x_s, st_s, gw.f32, gb.f32, x_s.shape[0], HW_, C, 8, 1e-5, True, y)
                    return x_s, st_s, y

                def up_rows(x_rows, f, H_, W_, C):
                    n = x_rows.shape[0]
                    up = torch.empty(n, 2 * H_, 2 * W_, C, dtype=BF16, device=dev)
                    if sel["rows"] is None:
                        k.upsample_add(x_rows, f, n, Q, H_, W_, C, up)
                    else:
                        k.upsample_add_rows(x_rows, f, sel["rows"], n, Q, H_, W_, C, up)
                    return up

                def conv_grads(dpre, x_in, Wp, bp, hw):
                    Co = dpre.shape[-1]
                    if Wp.g is not None:
                        tmp = ops.conv2d_wgrad(dpre, x_in, Wp.w.shape, pad=1)
                        _add_conv_grad(Wp.g, tmp)
                        if bp.g is not None:
                            ops.bias_grad(dpre.view(-1, Co), out=bp.g)
                    return ops.conv2d_dgrad(dpre, Wp.w, hw, pad=1)

                def gn_grads(g, x_s, st_s, gw, gb, HW_, C):
                    n = g.shape[0]
                    dpre = torch.empty_like(g)
                    bstats = torch.empty(n, 8, 2, dtype=torch.float32, device=dev)
                    k.groupnorm_bwd(g, None, x_s, st_s, gw.f32, n, HW_, C, 8, 1e-5, True, dpre, gw.g, gb.g if gw.g is not None else None, bstats, beta=gb.f32)
                    return dpre

                a4v = engine.Var(pre4)     # gradient slots of the (never materialised) normalised activations
                def tail_bwd():
                    g = mv.take_grad()
                    if g is None:
                        return
                    g = matched_rows_of(g)
                    # out_lay
                    x5, s5, a5r = norm_rows(pre5, st5, g5w, g5b, H5 * W5_, C5)
                    g8 = out_lay_grads(g, a5r, Wo, bo, wo8)
                    da5 = ops.conv2d_dgrad(g8, wo8, (H5, W5_), pad=1)
                    del a5r, g8
                    # gn5 + lay5 on (adapter3(f2) + up2(relu(gn4(pre4))))
                    dpre5 = gn_grads(da5, x5, s5, g5w, g5b, H5 * W5_, C5)
                    x4, s4, a4r = norm_rows(pre4, st4, g4w, g4b, H4 * W4_, C4)
                    up5 = up_rows(a4r, fb, H4, W4_, C4)
                    dup5 = conv_grads(dpre5, up5, W5, b5, (H5, W5_))
                    del up5, a4r, dpre5, da5
                    up_grads(dup5, a4v, f2, 3, wa3, H4, W4_)
                    del dup5
                    # gn4 + lay4 on (adapter2(f3) + up2(a3))
                    dpre4 = gn_grads(a4v.take_grad(), x4, s4, g4w, g4b, H4 * W4_, C4)
                    up4 = up_rows(pick(a3.data), fa, 2 * h, 2 * w, C3)
                    dup4 = conv_grads(dpre4, up4, W4, b4, (H4, W4_))
                    del up4, dpre4
                    up_grads(dup4, a3, f3, 2, wa2, 2 * h, 2 * w)

                tape.record(tail_bwd)
                return [mv], None

            a4 = fpn_stage(a3, f3, 2, h4, w4)
            a5 = fpn_stage(a4, f2, 3, h3, w3)
            bo8 = torch.zeros(8, dtype=torch.float32, device=dev)
            bo8[:1] = bo.f32
            o8 = ops.conv2d(a5.data, wo8, pad=1, shift=bo8)                         # [BQ,Hm,Wm,8] bf16
            masks = o8[..., 0].float().view(B, Q, Hm, Wm).contiguous()
            mv = engine.Var(masks)
            mv.raw_grad = True

            def out_bwd():
                g = mv.take_grad()
                if g is None:
                    return
                g = matched_rows_of(g)
                g8 = out_lay_grads(g, pick(a5.data), Wo, bo, wo8)
                a5.grad = ops.conv2d_dgrad(g8, wo8, (Hm, Wm), pad=1, res=a5.grad)

            tape.record(out_bwd)
            return [mv], None

        sink = _MatchedRows()
        (masks,) = functions.run_program(prog, named, [hs_last, memory, src_proj, c4, c3, c2], cache=self._cache, training=self.training,
                                         transforms=transforms)
        if MATCHED_ONLY_BACKWARD and masks.requires_grad:
            masks.toist_matched_rows = sink        # read by mask_losses / mask_losses_static
        return masks

    # ---- reference-compatible forward ----------------------------------------------------------------------
    def forward(self, samples: NestedTensor, captions, encode_and_save=True, memory_cache=None):
        if encode_and_save:
            assert memory_cache is None
            if not isinstance(samples, NestedTensor):
                samples = NestedTensor.from_tensor_list(samples)
            mc = self.detr.encode(samples, captions, levels=(1, 2, 3, 4))
            nat = mc["_native"]
            # reference keys (segmentation.py:76-78): NCHW fp32 views for API users
            mc["features_4_mask"] = [NestedTensor(f.permute(0, 3, 1, 2), nearest_mask(samples.mask, f.shape[1:3])) for f in nat["features"]]
            B, HW, d = nat["src_proj"].shape
            h, w = nat["features"][-1].shape[1:3]
            mc["src_proj_4_mask"] = nat["src_proj"].view(B, h, w, d).permute(0, 3, 1, 2)
            return mc
        assert memory_cache is not None
        out = self.detr.decode(memory_cache)
        nat = memory_cache["_native"]
        c2, c3, c4, c5 = nat["features"]
        B, h, w, _ = c5.shape
        HW = h * w
        stack = out["_stacked"]["hs"]                                   # [L, B*Q, d] bf16
        Q = stack.shape[1] // B
        d = stack.shape[-1]
        mem = nat["memory"].view(B, nat["S"], d)[:, :HW].reshape(B * HW, d)
        src = nat["src_proj"].reshape(B * HW, d)
        out["pred_masks"] = self._masks(stack[-1], mem, src, c4, c3, c2, nat["feat_mask"], B, Q, h, w)
        return out


# ------------------------------------------------------------------------------------------ mask losses
class _MaskLossFn(torch.autograd.Function):
    """(loss_mask, loss_dice) of SetCriterion.loss_masks for the matched (prediction, target) pairs."""

    @staticmethod
    def forward(ctx, pred, pred_row, gt, gt_row, num_boxes, TH, TW, sink=None, seg=None, valid=None):
        """valid: optional device int32 [4] = {VH, VW, hs, ws} -- the batch's own padded mask size inside a [TH, TW] bucket and the prediction size
        the reference would have had for it (matcher.StaticTargets.valid_hw, include/toist_hip.h): that corner of the prediction is resized onto
        that corner of the targets and loss_mask is the mean over VH * VW pixels, as mdetr.py:843-851 does on that batch."""
        T = pred_row.numel()
        ctx.sink, ctx.seg = sink, seg
        h, w = pred.shape[-2:]
        sums = torch.zeros(T, 4, dtype=torch.float32, device=pred.device)
        k.mask_loss_fwd(pred, pred_row, gt, gt_row, T, h, w, TH, TW, 0.25, sums, valid_hw=valid)
        area = torch.full((), float(TH * TW), device=pred.device) if valid is None else (valid[0] * valid[1]).float()
        focal = (sums[:, 0] / area).sum() / num_boxes
        dice = (1 - (2 * sums[:, 1] + 1) / (sums[:, 2] + sums[:, 3] + 1)).sum() / num_boxes
        ctx.save_for_backward(pred, pred_row, gt, gt_row, sums, num_boxes, area, valid)
        ctx.dims = (T, h, w, TH, TW)
        return torch.stack([focal, dice])

    @staticmethod
    def backward(ctx, g):
        pred, pred_row, gt, gt_row, sums, num_boxes, area, valid = ctx.saved_tensors
        T, h, w, TH, TW = ctx.dims
        coef = torch.stack([g[0] / (area * num_boxes), g[1] / num_boxes]).float().contiguous()
        sink = ctx.sink
        none = (None,) * 9
        if sink is not None and sink.grad is None:
            # matched maps only: pair t's gradient goes to row t of a [T,h,w] buffer that the mask head's backward picks up together with the
            # row indices; autograd carries a zero SENTINEL (one element, expanded) that the sink keeps alive -- see matched_rows_of
            rows_grad = torch.zeros(T, h, w, dtype=torch.float32, device=pred.device)
            k.mask_loss_bwd(pred, pred_row, gt, gt_row, T, h, w, TH, TW, 0.25, sums, coef, rows_grad, compact=True, valid_hw=valid)
            zero = _zero_sentinel(pred.device)
            sink.grad, sink.rows, sink.seg, sink.sentinel, sink.version = rows_grad, pred_row, ctx.seg, zero, zero._version
            return (zero.expand(pred.shape),) + none
        dpred = torch.zeros_like(pred)
        k.mask_loss_bwd(pred, pred_row, gt, gt_row, T, h, w, TH, TW, 0.25, sums, coef, dpred, valid_hw=valid)
        return (dpred,) + none


_OFFSETS = {}
_SEG = {}


def _match_offsets(counts, sizes, dev):
    """Per matched pair: image index and first target row of that image (device int64, cached per batch signature so
    a steady-state step does no host-to-device copy -- needed for hipGraph capture)."""
    key = (counts, sizes, str(dev))
    ent = _OFFSETS.get(key)
    if ent is None:
        if len(_OFFSETS) > 64:
            _OFFSETS.clear()
        b_idx = torch.cat([torch.full((c,), i, dtype=torch.int64) for i, c in enumerate(counts)]).to(dev)
        t_base = torch.cat([torch.full((c,), sum(sizes[:i]), dtype=torch.int64) for i, c in enumerate(counts)]).to(dev)
        ent = _OFFSETS[key] = (b_idx, t_base)
    return ent


_ARANGE = {}


def mask_losses_static(outputs, st, match, layer, L):
    """mask_losses on a matcher.StaticTargets image: the pair table has the fixed capacity st.cap; which slots are live, which image a
    pair belongs to and where layer `layer`'s pairs start (the matcher packs the layers with stride match_off[B]) are all computed ON THE
    DEVICE from st.match_off / st.tgt_off, so the launches are the same for every batch (hipGraph replay)."""
    pred = outputs["pred_masks"].float().contiguous()                      # [B,Q,hm,wm]
    B, Q = pred.shape[:2]
    dev = pred.device
    cap = st.cap
    if st.masks is None:
        raise ValueError("mask losses on StaticTargets need StaticTargets(mask_hw=(H, W))")
    TH, TW = st.mask_hw
    j = _ARANGE.get((cap, str(dev)))
    if j is None:
        j = _ARANGE[(cap, str(dev))] = torch.arange(cap, dtype=torch.int64, device=dev)
    mo = st.match_off.to(torch.int64)
    mtot = mo[B]
    live = j < mtot
    pos = (layer * mtot + j).clamp(max=L * cap - 1)
    src = match.src.reshape(-1).gather(0, pos)
    tgt = match.tgt.reshape(-1).gather(0, pos)
    img = torch.bucketize(j, mo[1:], right=True).clamp(max=B - 1)        # pair j belongs to the image whose run [match_off[i], match_off[i+1]) holds it
    pred_row = torch.where(live, img * Q + src, torch.full_like(src, -1)).to(torch.int32)
    gt_row = torch.where(live, st.tgt_off.to(torch.int64)[img] + tgt, torch.zeros_like(tgt)).to(torch.int32)
    sink = getattr(outputs["pred_masks"], "toist_matched_rows", None)
    seg = mo[:B + 1].to(torch.int32).contiguous() if sink is not None else None      # slots [match_off[i], match_off[i+1]) belong to image i
    vals = _MaskLossFn.apply(pred.view(B * Q, pred.shape[-2], pred.shape[-1]), pred_row.contiguous(), st.masks, gt_row.contiguous(), st.num_boxes.reshape(()).float(), TH, TW,
                             sink, seg, getattr(st, "valid_hw", None))
    return {"loss_mask": vals[0], "loss_dice": vals[1]}


def mask_losses(outputs, targets, match, layer, num_boxes):
    """SetCriterion.loss_masks (mdetr.py:827-853) on the device-resident assignment of `layer`."""
    pred = outputs["pred_masks"].float().contiguous()                      # [B,Q,hm,wm]
    B, Q = pred.shape[:2]
    dev = pred.device
    if match.src.shape[1] == 0:
        z = pred.sum() * 0
        return {"loss_mask": z, "loss_dice": z}
    TH = max(int(t["masks"].shape[-2]) for t in targets)
    TW = max(int(t["masks"].shape[-1]) for t in targets)
    rows = []
    for t in targets:                                                       # zero-padded like NestedTensor.from_tensor_list
        m = t["masks"].to(torch.uint8)
        if m.shape[-2:] != (TH, TW):
            m = torch.nn.functional.pad(m, (0, TW - m.shape[-1], 0, TH - m.shape[-2]))
        rows.append(m)
    gt = torch.cat(rows).contiguous()
    b_idx, t_base = _match_offsets(tuple(match.counts), tuple(match.sizes), dev)
    pred_row = (b_idx * Q + match.src[layer]).to(torch.int32)
    gt_row = (t_base + match.tgt[layer]).to(torch.int32)
    nb = num_boxes.reshape(()).float() if torch.is_tensor(num_boxes) else torch.tensor(float(num_boxes), device=dev)
    sink = getattr(outputs["pred_masks"], "toist_matched_rows", None)
    seg = None
    if sink is not None:
        counts = tuple(match.counts)
        seg = _SEG.get((counts, str(dev)))
        if seg is None:
            if len(_SEG) > 64:
                _SEG.clear()
            seg = _SEG[(counts, str(dev))] = torch.tensor([sum(counts[:i]) for i in range(len(counts) + 1)], dtype=torch.int32, device=dev)
    vals = _MaskLossFn.apply(pred.view(B * Q, pred.shape[-2], pred.shape[-1]), pred_row.contiguous(), gt, gt_row.contiguous(), nb, TH, TW, sink, seg)
    return {"loss_mask": vals[0], "loss_dice": vals[1]}
