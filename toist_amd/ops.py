"""Functional layer over the HIP kernels: linear / convolution / attention building blocks.

Shapes follow the kernels' native layouts (bf16, feature axis innermost): token matrices are
[rows, features], feature maps are NHWC, convolution weights are [Cout, R, S, Cin] ("KRSC", the
physical layout of a channels_last OIHW parameter).  Nothing here touches autograd; the modules in
toist_amd build their forward/backward passes out of these calls.
"""
import torch

from . import kernels as k
from .knobs import knob

BF16 = torch.bfloat16


def _ld(t):
    """Leading dimension of a 2-D (possibly column-sliced) row-major view."""
    assert t.dim() == 2 and t.stride(1) == 1, "need a row-major 2-D view"
    return t.stride(0)


SPLIT_TARGET = knob("TOIST_SPLIT_TARGET", 768)


def _split_k_for(tiles, ktiles, target=None, max_split=256):
    """k-slices for a reduction-heavy GEMM with few output tiles (wgrad): fill ~3 workgroups per CU
    while leaving every slice at least 4 k-tiles."""
    target = target or SPLIT_TARGET
    if tiles >= target or ktiles <= 8:
        return 1
    s = min(max_split, max(1, target // max(tiles, 1)), max(1, ktiles // 4))
    return max(1, s)


# ------------------------------------------------------------------------------------------ linear
SPLIT_LINEAR = knob("TOIST_SPLIT_LINEAR", True)
SPLIT_MAX_TILES, SPLIT_MIN_KTILES, SPLIT_KTILES_PER_SLICE, SPLIT_TARGET_WGS = 64, 32, 8, 384


def _split_for_linear(M, N, K):
    """k-slices for an nn.Linear GEMM with few 64x64 output tiles and a deep reduction (RoBERTa's FFN2 on 8 x 16 tokens: 24 tiles on
    256 CUs, each workgroup walking 48 k-tiles alone; the decoder's FFN2 on 800 queries: 52 tiles).  The slices are folded by
    csrc/gemm.hip splitk_epilogue_kernel, which applies the complete epilogue.  Measured (tools/dbg/gemm_splitk.py, us per launch,
    1 slice -> best): 128x768x3072 21.2 -> 11.8 (6 slices), its data gradient 26.3 -> 12.7; 128x768x2304 17.4 -> 11.1 (4);
    800x256x2048 15.9 -> 11.7 (4); K = 768 gains < 1 us and 208 tiles (3328x256x2048) gain nothing: left alone."""
    tiles = ((M + 63) // 64) * ((N + 63) // 64)
    ktiles = (K + 63) // 64
    if not SPLIT_LINEAR or tiles > SPLIT_MAX_TILES or ktiles < SPLIT_MIN_KTILES or N % 4:
        return 1
    return max(1, min(ktiles // SPLIT_KTILES_PER_SLICE, SPLIT_TARGET_WGS // tiles))


def linear(x, w, bias=None, *, out=None, out_dtype=BF16, act=k.ACT_NONE, res=None, alpha=1.0, pre_out=None, scale=None,
           drop_where=0, drop_p=0.0, drop_seed=0, tile=0, flags=0, split_k=None):
    """out[M,N] = act(alpha * x[M,K] @ w[N,K]^T * scale + bias (+dropout) + res)   (nn.Linear forward)."""
    M, K = x.shape
    N = w.shape[0]
    assert w.shape[1] == K
    if out is None:
        out = torch.empty(M, N, dtype=out_dtype, device=x.device)
    if split_k is None:
        split_k = _split_for_linear(M, N, K)
    k.gemm(M, N, K, k.A_ROWK, k.operand(x, _ld(x)), k.B_ROWK, k.operand(w, _ld(w)), out, _ld(out), alpha=alpha, scale=scale,
           shift=bias, res=res, ldr=_ld(res) if res is not None else 0, act=act, pre_out=pre_out, drop_where=drop_where,
           drop_p=drop_p, drop_seed=drop_seed, tile=tile, flags=flags, flops=2 * M * N * K, split_k=split_k, split_epilogue=split_k > 1)
    return out


def linear_dgrad(dy, w, *, out=None, res=None, act=k.ACT_NONE, aux=None, alpha=1.0, scale=None, flags=0, split_k=None):
    """dx[M,K] = (dy[M,N] @ w[N,K]) (+res) ; w is read k-major (no transposed copy)."""
    M, N = dy.shape
    K = w.shape[1]
    assert w.shape[0] == N
    if out is None:
        out = torch.empty(M, K, dtype=BF16, device=dy.device)
    if split_k is None:
        split_k = _split_for_linear(M, K, N)
    k.gemm(M, K, N, k.A_ROWK, k.operand(dy, _ld(dy)), k.B_KROW, k.operand(w, _ld(w)), out, _ld(out), alpha=alpha, scale=scale,
           res=res, ldr=_ld(res) if res is not None else 0, act=act, aux=aux, ldaux=_ld(aux) if aux is not None else 0,
           flags=flags, flops=2 * M * N * K, split_k=split_k, split_epilogue=split_k > 1)
    return out


def linear_wgrad(dy, x, *, out=None, alpha=1.0, flags=0, split_k=None, bias_out=None, defer=False, accumulate=True):
    """dw[N,K] (f32) += dy[M,N]^T @ x[M,K]; `out` must be zero-initialised or hold a running sum -- or, with accumulate=False, is
    simply overwritten (no zero fill, no read-modify-write).  bias_out (f32 [N]) additionally receives += sum_m dy[m, :]."""
    M, N = dy.shape
    K = x.shape[1]
    assert x.shape[0] == M
    if out is None:
        out = torch.zeros(N, K, dtype=torch.float32, device=dy.device)
    if split_k is None:
        tiles = ((N + 63) // 64) * ((K + 63) // 64)
        split_k = _split_k_for(tiles, (M + 63) // 64)
    k.gemm(N, K, M, k.A_KROW, k.operand(dy, _ld(dy)), k.B_KROW, k.operand(x, _ld(x)), out, _ld(out), alpha=alpha,
           accumulate=accumulate, split_k=split_k, flags=flags, flops=2 * M * N * K, a_colsum=bias_out, defer_reduce=defer)
    return out


def linear_wgrad_group(items, accumulate=True):
    """Weight (and bias) gradients of several nn.Linear layers of one shape -- the six layers of an encoder / decoder stack -- as one
    grouped launch: items = [(dy, x, out, bias_out)].  Separately each is a handful of tiles with a deep reduction (split-K + fold)."""
    dy0, x0, out0, b0 = items[0]
    M, N = dy0.shape
    K = x0.shape[1]
    if len(items) < 2 or len(items) > k.GROUP_MAX or any((it[3] is None) != (b0 is None) for it in items):
        for dy, x, out, bo in items:
            linear_wgrad(dy, x, out=out, bias_out=bo, defer=True, accumulate=accumulate)
        return
    rows = []
    for dy, x, out, bo in items:
        assert dy.shape == dy0.shape and x.shape == x0.shape and _ld(dy) == _ld(dy0) and _ld(x) == _ld(x0) and _ld(out) == _ld(out0)
        c_off, b_off = out.data_ptr() - out0.data_ptr(), (bo.data_ptr() - b0.data_ptr()) if bo is not None else 0
        assert c_off % 4 == 0 and b_off % 4 == 0 and out.dtype == torch.float32
        rows.append([dy.data_ptr(), x.data_ptr(), c_off // 4, 0, b_off // 4])
    table = k.group_table(rows, dy0.device)
    k.gemm(N, K, M, k.A_KROW, k.operand(dy0, _ld(dy0)), k.B_KROW, k.operand(x0, _ld(x0)), out0, _ld(out0), accumulate=accumulate, split_k=1,
           flops=2 * M * N * K * len(items), a_colsum=b0, batch=len(items), group=table)


def bias_grad(dy, out=None):
    M, N = dy.shape
    if out is None:
        out = torch.zeros(N, dtype=torch.float32, device=dy.device)
    k.colsum(dy, M, N, _ld(dy), out)
    return out


# ------------------------------------------------------------------------------------------ convolution
def conv_out_hw(H, W, R, S, stride, pad, dil=1):
    return (H + 2 * pad - dil * (R - 1) - 1) // stride + 1, (W + 2 * pad - dil * (S - 1) - 1) // stride + 1


def _small_conv_ok(R, S, stride, pad, dil, c_src, c_out, pixels):
    """csrc/smallconv.hip covers 3x3 / stride 1 / pad 1 with 8 / 16 / 32 source channels and <= 32 outputs; it pays
    where the tiled implicit GEMM is mostly padding: large pixel counts (measured at 160x160 x 800 maps)."""
    return R == 3 and S == 3 and stride == 1 and pad == 1 and dil == 1 and c_src in (8, 16, 32) and 0 < c_out <= 32 and c_out % 4 == 0 \
        and pixels >= (1 << 16)


def conv2d(x, w, *, stride=1, pad=0, dil=1, scale=None, shift=None, res=None, act=k.ACT_NONE, out=None, tile=0, res_bcast=None,
           out_dtype=BF16, cin_real=None):
    """NHWC implicit-GEMM convolution: x [N,H,W,C], w [Co,R,S,C] -> [N,OH,OW,Co] with the
    FrozenBatchNorm scale/shift (+residual, +ReLU) fused into the epilogue."""
    Nb, H, W, C = x.shape
    Co, R, S, Cw = w.shape
    assert Cw == C and x.is_contiguous() and w.is_contiguous()
    OH, OW = conv_out_hw(H, W, R, S, stride, pad, dil)
    if out is None:
        out = torch.empty(Nb, OH, OW, Co, dtype=out_dtype, device=x.device)
    if tile == 0 and _small_conv_ok(R, S, stride, pad, dil, C, Co, Nb * H * W) and scale is None and act == k.ACT_NONE and res_bcast is None \
            and out.dtype == BF16 and out.is_contiguous() and (res is None or (res.is_contiguous() and res.numel() == out.numel())):
        k.conv3x3_small(False, x, w, shift, res, out, Nb, H, W, C, Co)   # HBM-bound few-channel stage: direct kernel
        return out
    M = Nb * OH * OW
    Kred = R * S * C
    if R == 1 and S == 1 and stride == 1 and pad == 0:
        a_kind, a = k.A_ROWK, k.operand(x, C)
    else:
        a_kind = k.A_CONV
        a = k.operand(x, 0, geom=k.ConvGeom(H, W, C, OH, OW, R, S, stride, pad, dil))
        Kred = ((Kred + 31) // 32) * 32  # taps beyond R*S are zero-filled by the gather
    wk = w.view(Co, R * S * C)
    if Kred != R * S * C:  # stem only (C = 8): pad the reduction axis of the weights with zeros
        wk = torch.nn.functional.pad(wk, (0, Kred - R * S * C))
    k.gemm(M, Co, Kred, a_kind, a, k.B_ROWK, k.operand(wk, Kred), out, Co, scale=scale, shift=shift, res=res,
           ldr=Co if res is not None else 0, act=act, tile=tile, flops=2 * M * Co * R * S * (cin_real or C), res_bcast=res_bcast)
    return out


def conv2d_dgrad(dy, w, in_hw, *, stride=1, pad=0, dil=1, scale=None, res=None, act=k.ACT_NONE, aux=None, out=None, flags=0, tile=0):
    """dx [N,H,W,C] = transposed-gather of dy [N,OH,OW,Co] with w [Co,R,S,C] read in place (k-major)."""
    Nb, OH, OW, Co = dy.shape
    Cw, R, S, C = w.shape
    H, W = in_hw
    assert Cw == Co and dy.is_contiguous() and w.is_contiguous()
    if out is None:
        out = torch.empty(Nb, H, W, C, dtype=BF16, device=dy.device)
    if (OH, OW) == (H, W) and _small_conv_ok(R, S, stride, pad, dil, Co, C, Nb * H * W) and scale is None and act == k.ACT_NONE and aux is None \
            and out.is_contiguous() and (res is None or (res.is_contiguous() and res.numel() == out.numel())):
        k.conv3x3_small(True, dy, w, None, res, out, Nb, H, W, Co, C)
        return out
    M = Nb * H * W
    ldr = C if res is not None else 0
    ldaux = C if aux is not None else 0
    fl = 2 * Nb * OH * OW * Co * R * S * C  # algorithmic: the forward conv's MACs
    if R == 1 and S == 1 and stride == 1 and pad == 0:
        k.gemm(M, C, Co, k.A_ROWK, k.operand(dy, Co), k.B_KROW, k.operand(w.view(Co, C), C), out, C, scale=scale, res=res, ldr=ldr,
               act=act, aux=aux, ldaux=ldaux, flags=flags, flops=fl)
    elif R == 1 and S == 1 and pad == 0:
        # strided 1x1 (downsample): only rows (n, oy*stride, ox*stride) of dx receive a value
        k.gemm(Nb * OH * OW, C, Co, k.A_ROWK, k.operand(dy, Co), k.B_KROW, k.operand(w.view(Co, C), C), out, C, scale=scale,
               res=res, ldr=ldr, act=act, aux=aux, ldaux=ldaux, cmap=(H, W, OH, OW, stride), flags=flags, flops=fl)
    elif PARITY_DGRAD and R == 3 and S == 3 and stride == 2 and pad == 1 and dil == 1 and Co % 8 == 0:
        _dgrad3x3_s2(dy, w, out, scale, res, act, aux, flags, fl)
    else:
        a = k.operand(dy, 0, geom=k.ConvGeom(OH, OW, Co, H, W, R, S, stride, pad, dil))
        b = k.operand(w, R * S * C, kin=Co, tap_stride=C)
        k.gemm(M, C, R * S * Co, k.A_CONVT, a, k.B_KROW, b, out, C, scale=scale, res=res, ldr=ldr, act=act, aux=aux,
               ldaux=ldaux, flags=flags, flops=fl, tile=tile)
    return out


PARITY_DGRAD = knob("TOIST_PARITY_DGRAD", True)
# taps (r*3 + s) of a 3x3 / stride 2 / pad 1 kernel grouped by the parity (y & 1, x & 1) of the dx pixel they reach, each group in
# the order a plain stride-1 gather dy[yy + r', xx + s'] visits them:  (0,0): 1 tap | (0,1): 2 | (1,0): 2 | (1,1): 4
_S2_TAPS = (4, 5, 3, 7, 1, 8, 6, 2, 0)
_S2_CLASS = ((0, 0, 1, 1, 0), (0, 1, 1, 2, 1), (1, 0, 2, 1, 3), (1, 1, 2, 2, 5))       # (py, px, R', S', first tap)
_S2_PERM = {}


def _dgrad3x3_s2(dy, w, out, scale, res, act, aux, flags, fl):
    """Data gradient of a 3x3 / stride 2 / pad 1 convolution as four stride-1 gathers, one per parity class of the dx pixel.

    dx[n, y, x] sums dy[n, (y+1-r)/2, (x+1-s)/2] w[:, r, s] over the taps with even y+1-r and x+1-s: a quarter of the nine on
    average, so the transposed gather over all nine taps (the generic A_CONVT path) streams and multiplies 4x zeros.  Pixels with
    y = 2yy + py see r = 1 (py = 0) or r in {2, 0} (py = 1) at dy rows yy, yy + 1 -- a 1- or 2-tap stride-1 gather over the dy
    plane; likewise in x.  Each class is one implicit GEMM (A_CONV gather of dy, k-major weights with the class's taps made
    adjacent by one permuted copy of w) whose output rows scatter to (2yy + py, 2xx + px) through the epilogue's row map."""
    Nb, OH, OW, Co = dy.shape
    _, H, W, C = out.shape
    perm = _S2_PERM.get(dy.device)
    if perm is None:
        perm = _S2_PERM[dy.device] = torch.tensor(_S2_TAPS, dtype=torch.int64, device=dy.device)
    wp = w.view(Co, 9, C).index_select(1, perm)
    flat = lambda t, off: None if t is None else t.view(-1, C)[off:]
    for py, px, Rc, Sc, t0 in _S2_CLASS:
        PH, PW = (H - py + 1) // 2, (W - px + 1) // 2
        if PH == 0 or PW == 0:
            continue
        off = py * W + px
        a = k.operand(dy, 0, geom=k.ConvGeom(OH, OW, Co, PH, PW, Rc, Sc, 1, 0, 1))
        b = k.operand(wp.view(Co, 9 * C)[:, t0 * C:], 9 * C, kin=Co, tap_stride=C)
        k.gemm(Nb * PH * PW, C, Rc * Sc * Co, k.A_CONV, a, k.B_KROW, b, flat(out, off), C, scale=scale, res=flat(res, off),
               ldr=C if res is not None else 0, act=act, aux=flat(aux, off), ldaux=C if aux is not None else 0, cmap=(H, W, PH, PW, 2),
               flags=flags, flops=fl * Rc * Sc // 9)


def conv2d_wgrad(dy, x, w_shape, *, stride=1, pad=0, dil=1, out=None, flags=0, split_k=None, rscale=None, defer=False, accumulate=True):
    """dw [Co,R,S,C] (f32, accumulated) = sum over output pixels of dy (x) gathered x."""
    Nb, OH, OW, Co = dy.shape
    _, H, W, C = x.shape
    Cw, R, S, Cc = w_shape
    assert Cw == Co and Cc == C and dy.is_contiguous() and x.is_contiguous()
    if out is None:
        out = torch.zeros(Co, R, S, C, dtype=torch.float32, device=dy.device)
    if (OH, OW) == (H, W) and R == 3 and S == 3 and stride == 1 and pad == 1 and dil == 1 and C in (16, 32) and Co in (8, 16) and rscale is None \
            and split_k is None and Nb * H * W >= (1 << 16) and out.is_contiguous():
        k.wgrad3x3_small(dy, x, out, defer=defer, accumulate=accumulate)     # few-channel stage: direct kernel + batched fold
        return out
    P = Nb * OH * OW
    Nn = R * S * C
    if split_k is None:
        tiles = ((Co + 63) // 64) * ((Nn + 63) // 64)
        split_k = _split_k_for(tiles, (P + 63) // 64)
    a = k.operand(dy, Co)
    if R == 1 and S == 1 and stride == 1 and pad == 0:
        b_kind, b = k.B_KROW, k.operand(x, C)
    else:
        b_kind, b = k.B_CONVX, k.operand(x, 0, geom=k.ConvGeom(H, W, C, OH, OW, R, S, stride, pad, dil))
    k.gemm(Co, Nn, P, k.A_KROW, a, b_kind, b, out, Nn, accumulate=accumulate, split_k=split_k, flags=flags, rscale=rscale,
           flops=2 * P * Co * Nn, defer_reduce=defer)
    return out


GROUP_TILE = knob("TOIST_GROUP_TILE", 0)   # tile code of grouped weight-gradient launches (0 = the dispatcher's choice)
GROUP_TILE_3X3 = knob("TOIST_GROUP_TILE_3X3", 130)
GROUP_TILE_1X1 = knob("TOIST_GROUP_TILE_1X1", 134)
GROUP_SPLIT_TILE = knob("TOIST_GROUP_SPLIT_TILE", 130)   # grouped AND split along K: 128x64 tiles with twice the slices (+0.7% step over 64x64)
GROUP_MIN_TILES = 512   # below this many 64x64 output tiles in total the problems stay separate (they need split-K)


def conv2d_wgrad_group(items, w_shape, *, stride=1, pad=0, dil=1, accumulate=True):
    """Weight gradients of several convolutions of ONE shape (the identical residual blocks of a stage) as one grouped GEMM
    launch: items = [(dy, x, out, rscale)], out f32 [Co,R,S,C] accumulated.  Separately each is 64-144 output tiles with a
    12800-51200 deep reduction, i.e. split along K plus a fold pass; together they fill the chip unsplit."""
    dy0, x0, out0, rs0 = items[0]
    Nb, OH, OW, Co = dy0.shape
    _, H, W, C = x0.shape
    Cw, R, S, Cc = w_shape
    Nn, P = R * S * C, Nb * OH * OW
    tiles = ((Co + 63) // 64) * ((Nn + 63) // 64) * len(items)
    if len(items) < 2 or len(items) > k.GROUP_MAX or any((it[3] is None) != (rs0 is None) for it in items):
        for dy, x, out, rs in items:
            conv2d_wgrad(dy, x, w_shape, stride=stride, pad=pad, dil=dil, out=out, rscale=rs, defer=True, accumulate=accumulate)
        return
    # too few tiles to fill the chip even together: the group is also split along K (partials folded by the batched reduction)
    split_k = _split_k_for(tiles, (P + 63) // 64) if tiles < GROUP_MIN_TILES else 1
    rows = []
    for dy, x, out, rs in items:
        assert dy.shape == dy0.shape and x.shape == x0.shape and dy.is_contiguous() and x.is_contiguous() and out.is_contiguous()
        c_off, r_off = out.data_ptr() - out0.data_ptr(), (rs.data_ptr() - rs0.data_ptr()) if rs is not None else 0
        assert c_off % 4 == 0 and r_off % 4 == 0 and out.dtype == torch.float32 and (rs is None or rs.dtype == torch.float32)
        rows.append([dy.data_ptr(), x.data_ptr(), c_off // 4, r_off // 4, 0])
    table = k.group_table(rows, dy0.device)
    a = k.operand(dy0, Co)
    if R == 1 and S == 1 and stride == 1 and pad == 0:
        b_kind, b = k.B_KROW, k.operand(x0, C)
    else:
        b_kind, b = k.B_CONVX, k.operand(x0, 0, geom=k.ConvGeom(H, W, C, OH, OW, R, S, stride, pad, dil))
    tile = GROUP_TILE
    if tile == 0 and split_k == 1:
        # measured on the 22 grouped layer-3 problems (K = 12800, tools/run_timeline.sh): 3x3 714 / 562 / 598 us and 1x1 913 / 886 / 818 us
        # with 64x64 / 128x128 / 128x64 tiles -- deep reductions with enough tiles to fill the chip want the larger tiles.  Re-measured with the
        # 64x128 tile (code 134) on whole steps, same box: (3x3, 1x1) = (129, 130) 517-519 images/s, (134, 134) 522-526, (130, 134) 523-527
        big = GROUP_TILE_3X3 if R * S > 1 else GROUP_TILE_1X1
        n_big = ((Co + (63 if big == 134 else 127)) // (64 if big == 134 else 128)) * ((Nn + (63 if big == 130 else 127)) // (64 if big == 130 else 128)) * len(items)
        tile = big if n_big >= 256 else 0
    elif tile == 0 and GROUP_SPLIT_TILE:
        tile = GROUP_SPLIT_TILE
        split_k = min(2 * split_k, max(1, (P + 63) // 64 // 4))
    k.gemm(Co, Nn, P, k.A_KROW, a, b_kind, b, out0, Nn, accumulate=accumulate, split_k=split_k, rscale=rs0, batch=len(items), tile=tile,
           flops=2 * P * Co * Nn * len(items), group=table, group_out=[(it[2], it[3]) for it in items])


# ------------------------------------------------------------------------------------------ attention
def round8(n):
    return (n + 7) // 8 * 8


def attn_scores(q, kmat, B, H, Sq, Sk, dh, scale, out=None):
    """scores[b,h,i,j] = scale * q[b,i,h,:] . k[b,j,h,:]; q/k are column slices of packed projection
    buffers: row (b*S + s), feature (h*dh + e).  Output [B*H, Sq, round8(Sk)] bf16."""
    ld = round8(Sk)
    if out is None:
        out = torch.empty(B * H, Sq, ld, dtype=BF16, device=q.device)
    k.gemm(Sq, Sk, dh, k.A_ROWK, k.operand(q, _ld(q), bs_outer=Sq * _ld(q), bs_inner=dh), k.B_ROWK,
           k.operand(kmat, _ld(kmat), bs_outer=Sk * _ld(kmat), bs_inner=dh), out, ld, batch=B * H, batch_inner=H,
           cs_outer=H * Sq * ld, cs_inner=Sq * ld, alpha=scale, tile=64, flops=2 * B * H * Sq * Sk * dh)
    return out


def attn_context(p, v, B, H, Sq, Sk, dh, out):
    """ctx[b,i,h,:] = sum_j p[b,h,i,j] * v[b,j,h,:]; `out` is a [B*Sq, H*dh] (slice of a) buffer."""
    ld = p.shape[-1]
    k.gemm(Sq, dh, Sk, k.A_ROWK, k.operand(p, ld, bs_outer=H * Sq * ld, bs_inner=Sq * ld), k.B_KROW,
           k.operand(v, _ld(v), bs_outer=Sk * _ld(v), bs_inner=dh), out, _ld(out), batch=B * H, batch_inner=H,
           cs_outer=Sq * _ld(out), cs_inner=dh, tile=64, flops=2 * B * H * Sq * Sk * dh)
    return out


def attn_backward(p_used, ds_scale, q, kmat, v, dctx, B, H, Sq, Sk, dh, dq, dk, dv, softmax_bwd_fn):
    """Backward of softmax(scale*q.k^T).v for packed per-head slices.

    p_used: probabilities that multiplied v in the forward pass (after dropout when training).
    softmax_bwd_fn(dp) -> ds maps dP (wrt p_used) to dS (wrt the scaled scores)."""
    ld = p_used.shape[-1]
    dev = q.device
    # dV[b,j,h,:] = sum_i p[b,h,i,j] * dctx[b,i,h,:]
    k.gemm(Sk, dh, Sq, k.A_KROW, k.operand(p_used, ld, bs_outer=H * Sq * ld, bs_inner=Sq * ld), k.B_KROW,
           k.operand(dctx, _ld(dctx), bs_outer=Sq * _ld(dctx), bs_inner=dh), dv, _ld(dv), batch=B * H, batch_inner=H,
           cs_outer=Sk * _ld(dv), cs_inner=dh, tile=64, flops=2 * B * H * Sq * Sk * dh)
    # dP[b,h,i,j] = dctx[b,i,h,:] . v[b,j,h,:]
    dp = torch.empty(B * H, Sq, ld, dtype=BF16, device=dev)
    k.gemm(Sq, Sk, dh, k.A_ROWK, k.operand(dctx, _ld(dctx), bs_outer=Sq * _ld(dctx), bs_inner=dh), k.B_ROWK,
           k.operand(v, _ld(v), bs_outer=Sk * _ld(v), bs_inner=dh), dp, ld, batch=B * H, batch_inner=H, cs_outer=H * Sq * ld,
           cs_inner=Sq * ld, tile=64, flops=2 * B * H * Sq * Sk * dh)
    ds = softmax_bwd_fn(dp)
    # dQ = scale * dS @ K ; dK = scale * dS^T @ Q
    k.gemm(Sq, dh, Sk, k.A_ROWK, k.operand(ds, ld, bs_outer=H * Sq * ld, bs_inner=Sq * ld), k.B_KROW,
           k.operand(kmat, _ld(kmat), bs_outer=Sk * _ld(kmat), bs_inner=dh), dq, _ld(dq), batch=B * H, batch_inner=H,
           cs_outer=Sq * _ld(dq), cs_inner=dh, alpha=ds_scale, tile=64, flops=2 * B * H * Sq * Sk * dh)
    k.gemm(Sk, dh, Sq, k.A_KROW, k.operand(ds, ld, bs_outer=H * Sq * ld, bs_inner=Sq * ld), k.B_KROW,
           k.operand(q, _ld(q), bs_outer=Sq * _ld(q), bs_inner=dh), dk, _ld(dk), batch=B * H, batch_inner=H,
           cs_outer=Sk * _ld(dk), cs_inner=dh, alpha=ds_scale, tile=64, flops=2 * B * H * Sq * Sk * dh)
