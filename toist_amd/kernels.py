"""Tensor-level launchers for the C-ABI kernels of libtoist_hip.so.

Every function takes torch tensors that already live on the HIP device, passes raw pointers + sizes
through ctypes on torch's current stream, and returns immediately (no host sync).  torch is used for
memory and streams only.  There is no fallback: a CPU tensor or a missing library raises.
"""
import ctypes
import os

import torch

from . import _lib
from .knobs import knob
from ._lib import (A_CONV, A_CONVT, A_KROW, A_ROWK, ACT_GELU, ACT_GELU_BWD, ACT_MASK_POS, ACT_NONE, ACT_RELU,
                   ACT_SIGMOID, ACT_SIGMOID_BWD, B_CONVX, B_KROW, B_ROWK, Epilogue, Gemm, Operand)

__all__ = [
    "A_ROWK", "A_KROW", "A_CONV", "A_CONVT", "B_ROWK", "B_KROW", "B_CONVX", "ACT_NONE", "ACT_RELU", "ACT_GELU",
    "ACT_SIGMOID", "ACT_MASK_POS", "ACT_GELU_BWD", "ACT_SIGMOID_BWD", "ConvGeom", "operand", "gemm", "matcher",
    "layernorm_fwd", "layernorm_bwd", "softmax_fwd", "softmax_bwd", "colsum", "add", "dropout", "pack_image", "maxpool3x3s2", "stem_fwd",
    "unpack_nhwc", "sine_position", "embed_fwd", "embed_bwd", "criterion_fwd", "criterion_bwd", "attnmap_softmax_fwd", "attnmap_softmax_bwd",
    "groupnorm_fwd", "groupnorm_apply", "groupnorm_bwd", "upsample_add", "upsample_add_bwd", "sum_queries", "sum_segments", "upsample_add_rows", "mask_stage_fwd", "mask_loss_fwd", "mask_loss_bwd",
]


# bench.py sets this to {"key": (tile, a_kind, b_kind) | set of such | predicate(key) | None, "records": [], "other": {}} to time GEMM
# launches with HIP events on the launch stream (roofline accounting); None = no instrumentation.
PROFILE = None
# Optional int64[1] device tensor added to every dropout seed inside the kernels.  A training loop that
# replays a captured hipGraph bumps it once per step so each replay draws fresh dropout masks.
SEED_DEV = None
FORCE_TILE = 0    # experiments: overrides tile == 0 (auto) in every gemm() call
FORCE_SPLIT = 0   # experiments: overrides split_k when > 0 and the call accumulates
DEBUG_FLAGS = 0   # experiments: extra flag bits (ablation switches of the PROF kernels) when DEBUG_WS is set
DEBUG_WS = None   # experiments: f32 device buffer handed to un-split gemm() calls as `workspace` with flags bit 11 (per-phase cycle counters of panel2_kernel)


_WS = {}


def _workspace(elems, device):
    """Caller-owned split-K scratch handed to the library (grown on demand, reused across calls)."""
    key = (device, _raw_stream())  # one scratch per stream: launches on different streams overlap
    ws = _WS.get(key)
    if ws is None or ws.numel() < elems:
        ws = torch.empty(max(elems, 1 << 22), dtype=torch.float32, device=device)
        _WS[key] = ws
    return ws


# ---- deferred split-K reductions -----------------------------------------------------------------------------
# A weight-gradient GEMM with few output tiles is cut into k-slices whose fp32 partials land in an arena; with
# defer_reduce=True the fold into the gradient buffer is not launched per GEMM (178 launches of ~6 us per step) but
# queued, and flush_reductions() folds everything queued on the current stream with one launch per 48 GEMMs.
# The queue must be flushed before anything reads those outputs: engine.Tape.backward() does it at program end.
GEMM_DEFER_REDUCE = 2
GEMM_SPLIT_EPILOGUE = 4
_ARENA = {}     # (device, stream) -> [buffer, used elements]
_PENDING = {}   # (device, stream) -> list of (ReduceDesc, keep-alive tensors)


def _arena_take(elems, device):
    key = (device, _raw_stream())
    ent = _ARENA.get(key)
    elems = (elems + 63) // 64 * 64
    if ent is None or ent[1] + elems > ent[0].numel():
        flush_reductions()                      # queued descriptors point into the old arena
        size = max(elems, 1 << 26 if ent is None else 2 * ent[0].numel())
        ent = _ARENA[key] = [torch.empty(size, dtype=torch.float32, device=device), 0]
    off = ent[1]
    ent[1] += elems
    return ent[0][off:off + elems]


def flush_reductions():
    """Fold every queued split-K partial of the current stream into its output (one launch per 48 GEMMs)."""
    if not torch.cuda.is_available():
        return
    key_s = _raw_stream()
    for key in [kk for kk in _PENDING if kk[1] == key_s]:
        items = _PENDING.pop(key)
        if items:
            arr = (_lib.ReduceDesc * len(items))(*[it[0] for it in items])
            _lib.check(_lib.lib().toist_splitk_reduce_batch(ctypes.cast(arr, ctypes.c_void_p), len(items), _stream()), "toist_splitk_reduce_batch")
        ent = _ARENA.get(key)
        if ent is not None:
            ent[1] = 0


GROUP_MAX = 64


_GROUP_TABLES = {}      # (device, rows) -> filled device table


def group_table(rows, device):
    """Device toist_group table from host rows [a_ptr, b_ptr, c_off, rscale_off, colsum_off] (csrc/gemm.hip: the rows travel as kernel
    arguments).  Tables are kept: a training loop presents the same pointers step after step (always under hipGraph replay, nearly
    always with the caching allocator), so the 25 fill launches of a step run once -- a table filled before a capture is simply read
    by the captured launches.  A table needed DURING a capture is filled beside the graph (see below)."""
    key = (str(device), tuple(int(v) for r in rows for v in (list(r) + [0] * (6 - len(r)))))
    hit = _GROUP_TABLES.get(key)
    if hit is not None:
        return hit
    n = len(rows)
    flat = (ctypes.c_int64 * (6 * n))(*key[1])
    if torch.cuda.is_current_stream_capturing() and FILL_TABLES_OUTSIDE_GRAPH:
        # The pointers in `rows` are fixed for the life of the graph being captured, so the table is filled NOW, by a launch on a
        # stream that is not capturing, instead of by a kernel node that would re-write the same 48 bytes per row on every replay
        # (25 launches per training step).  The table must NOT come from the graph's memory pool: a block handed out during a capture
        # may have held an earlier temporary of the same capture, whose producer kernels run again at every replay and would overwrite
        # a table that is no longer re-filled behind them (seen as a memory fault in the first replayed step).  It is carved from an
        # arena allocated before the capture; when the arena is exhausted the fill stays a graph node.
        dev = _arena_table(str(device), n)
        if dev is not None:
            side = _fill_stream(device)
            _lib.check(_lib.lib().toist_group_fill(ctypes.cast(flat, ctypes.c_void_p), n, _p(dev), ctypes.c_void_p(side.cuda_stream)), "toist_group_fill")
            return dev
    dev = torch.empty(n, 6, dtype=torch.int64, device=device)
    _lib.check(_lib.lib().toist_group_fill(ctypes.cast(flat, ctypes.c_void_p), n, _p(dev), _stream()), "toist_group_fill")
    if not torch.cuda.is_current_stream_capturing():
        if len(_GROUP_TABLES) >= 1024:
            _GROUP_TABLES.clear()
        _GROUP_TABLES[key] = dev
    return dev


# Opt-in (harness.CapturedTrainStep, bench.py): whoever captures sets this around the capture and calls sync_table_fills() before the first
# replay -- the fill is not ordered against the graph by any stream dependency (a wait on it cannot be recorded into the capture, and
# synchronising the fill stream during a capture invalidates the capture on ROCm 7.2).  Other captures keep the fill as a graph node.
FILL_TABLES_OUTSIDE_GRAPH = False
_TABLE_ARENAS = {}      # device -> [int64 tensor [rows, 6] allocated outside any capture, rows used]
_TABLE_ARENA_ROWS = 1 << 16     # 3 MB: ~2000 tables of 32 problems
_FILL_STREAMS = {}
_RETIRED_ARENAS = []


def _arena_table(dev_key, n):
    ar = _TABLE_ARENAS.get(dev_key)
    if ar is None or ar[1] + n > ar[0].shape[0]:
        return None
    t = ar[0][ar[1]:ar[1] + n]
    ar[1] += n
    return t


def _arena_prepare(device):
    """(outside captures) make sure the table arena of `device` has room for a capture's worth of tables"""
    key = str(device)
    ar = _TABLE_ARENAS.get(key)
    if ar is None or ar[1] + 4096 > ar[0].shape[0]:
        if ar is not None:
            _RETIRED_ARENAS.append(ar[0])       # captured graphs read their tables from it for as long as they live
        _TABLE_ARENAS[key] = [torch.empty(_TABLE_ARENA_ROWS, 6, dtype=torch.int64, device=device), 0]


def _fill_stream(device):
    s = _FILL_STREAMS.get(str(device))
    if s is None:
        s = torch.cuda.Stream(device=device)
        _FILL_STREAMS[str(device)] = s
    return s


def sync_table_fills():
    """Wait for table fills launched beside a capture (call once after capture_end, before the first replay)."""
    for s in _FILL_STREAMS.values():
        s.synchronize()


class tables_beside_graph:
    """with kernels.tables_beside_graph(): <capture(s)> -- pointer tables of grouped launches are filled once, beside the capture, and
    the fills are joined on exit (TOIST_FILL_OUTSIDE_GRAPH=0 keeps them inside the graph)."""

    def __enter__(self):
        global FILL_TABLES_OUTSIDE_GRAPH
        self.old = FILL_TABLES_OUTSIDE_GRAPH
        FILL_TABLES_OUTSIDE_GRAPH = knob("TOIST_FILL_OUTSIDE_GRAPH", True)
        if FILL_TABLES_OUTSIDE_GRAPH and not torch.cuda.is_current_stream_capturing():
            _arena_prepare(torch.device("cuda", torch.cuda.current_device()))
        return self

    def __exit__(self, *exc):
        global FILL_TABLES_OUTSIDE_GRAPH
        FILL_TABLES_OUTSIDE_GRAPH = self.old
        sync_table_fills()
        return False


def _raw_stream():
    """hipStream_t of torch's current stream as an int.  torch.cuda.current_stream() builds a Stream object through three
    Python layers (~9 us): at ~1500 launches per step that alone was 13 ms of host time; the C accessor takes ~0.3 us."""
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


def _stream():
    return ctypes.c_void_p(_raw_stream())


def _p(t, dtype=None):
    """Device pointer of a tensor (None -> NULL).  Refuses host tensors: no CPU path exists."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("toist_amd kernels need device tensors (got a CPU tensor); there is no CPU fallback")
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f"expected {dtype}, got {t.dtype}")
    return t.data_ptr()


class ConvGeom:
    """Gather geometry of an implicit-GEMM operand: source [N,SH,SW,SC] NHWC, pixel space [N,PH,PW]."""

    __slots__ = ("SH", "SW", "SC", "PH", "PW", "R", "S", "stride", "pad", "dil")

    def __init__(self, SH, SW, SC, PH, PW, R, S, stride, pad, dil=1):
        self.SH, self.SW, self.SC, self.PH, self.PW = SH, SW, SC, PH, PW
        self.R, self.S, self.stride, self.pad, self.dil = R, S, stride, pad, dil


def operand(t, ld=0, bs_outer=0, bs_inner=0, kin=0, tap_stride=0, geom=None):
    o = Operand()
    o.ptr = _p(t, torch.bfloat16)
    o.ld, o.bs_outer, o.bs_inner, o.kin, o.tap_stride = ld, bs_outer, bs_inner, kin, tap_stride
    if geom is not None:
        o.SH, o.SW, o.SC, o.PH, o.PW = geom.SH, geom.SW, geom.SC, geom.PH, geom.PW
        o.R, o.S, o.stride, o.pad, o.dil = geom.R, geom.S, geom.stride, geom.pad, geom.dil
    return o


def gemm(M, N, K, a_kind, a, b_kind, b, c, ldc, *, batch=1, batch_inner=1, cs_outer=0, cs_inner=0, split_k=1, tile=0,
         flags=0, alpha=1.0, scale=None, shift=None, rscale=None, res=None, ldr=0, aux=None, ldaux=0, act=ACT_NONE, pre_out=None,
         accumulate=False, cmap=None, drop_where=0, drop_p=0.0, drop_seed=0, flops=0, a_colsum=None, res_bcast=None,
         defer_reduce=False, group=None, group_out=None, a2=None, a2_from=0, split_epilogue=False):
    """C = epilogue(A @ B^T); see include/toist_hip.h.  `a`/`b` are Operand structs from operand().
    `flops` = algorithmic FLOPs of the call (bench.py's roofline accounting only)."""
    if tile == 0 and FORCE_TILE:
        tile = FORCE_TILE
    if FORCE_SPLIT and accumulate:
        split_k = FORCE_SPLIT
    d = Gemm()
    d.M, d.N, d.K, d.a_kind, d.b_kind, d.a, d.b = M, N, K, a_kind, b_kind, a, b
    if c.dtype == torch.float32:
        d.epi.out_f32 = 1
    elif c.dtype != torch.bfloat16:
        raise TypeError("gemm output must be bf16 or f32")
    d.c, d.ldc = _p(c), ldc
    d.cs_outer, d.cs_inner, d.batch, d.batch_inner = cs_outer, cs_inner, batch, batch_inner
    d.split_k, d.tile, d.flags = split_k, tile, flags
    e = d.epi
    e.alpha = alpha
    e.scale, e.shift = _p(scale, torch.float32), _p(shift, torch.float32)
    e.rscale = _p(rscale, torch.float32)
    e.res, e.ldr = _p(res, torch.bfloat16), ldr
    e.aux, e.ldaux = _p(aux, torch.bfloat16), ldaux
    e.act = act
    e.pre_out = _p(pre_out, torch.bfloat16)
    e.accumulate = 1 if accumulate else 0
    if cmap is not None:
        e.cmap = 1
        e.cH, e.cW, e.cOH, e.cOW, e.cst = cmap
    if res_bcast is not None:
        e.res_div, e.res_mod = res_bcast
    e.drop_where, e.drop_p, e.drop_seed = drop_where, drop_p, drop_seed
    e.drop_seed_dev = _p(SEED_DEV) if drop_where else None
    d.a_colsum = _p(a_colsum, torch.float32)
    d.group = _p(group, torch.int64)          # [batch, 6] int64 rows (a, b pointers; c, rscale, colsum, shift offsets): toist_group
    if a2 is not None:                        # output columns >= a2_from take their A rows from a2 (packed in_proj on two inputs)
        d.a2, d.a2_from = _p(a2, torch.bfloat16), a2_from
    deferred = None
    if split_k > 1 and split_epilogue:
        # k-slices folded by a second kernel that applies the complete epilogue (any output type): csrc/gemm.hip splitk_epilogue_kernel
        d.workspace = _p(_workspace(split_k * M * N, c.device), torch.float32)
        d.flags |= GEMM_SPLIT_EPILOGUE
    elif split_k > 1 and group is not None:
        # grouped + split: [problem][k-slice][M][N] partials, one queued fold per problem (group_out = [(c_i, rscale_i)])
        eff = int(_lib.lib().toist_gemm_effective_split(ctypes.byref(d)))
        if eff > 1:
            key = (c.device, _raw_stream())
            outs = {ci.data_ptr() for ci, _ in group_out}
            if any(it[0].out in outs for it in _PENDING.get(key, ())):
                flush_reductions()
            ws = _arena_take(eff * M * N * batch, c.device)
            d.workspace = _p(ws, torch.float32)
            d.flags |= GEMM_DEFER_REDUCE
            for i, (ci, ri) in enumerate(group_out):
                rd = _lib.ReduceDesc(ws.data_ptr() + 4 * i * eff * M * N, ci.data_ptr(), _p(ri, torch.float32), eff, M, N, ldc, alpha, 1 if accumulate else 0)
                _PENDING.setdefault(key, []).append((rd, (ci, ri)))
        else:
            d.split_k = 1
    elif split_k > 1:
        eff = int(_lib.lib().toist_gemm_effective_split(ctypes.byref(d))) if defer_reduce else 0
        if eff > 1:
            # a second deferred reduction into the same output would race with the queued one: fold first
            key = (c.device, _raw_stream())
            if any(it[0].out == c.data_ptr() for it in _PENDING.get(key, ())):
                flush_reductions()
            ws = _arena_take(eff * M * N, c.device)
            d.workspace = _p(ws, torch.float32)
            d.flags |= GEMM_DEFER_REDUCE
            rd = _lib.ReduceDesc(ws.data_ptr(), c.data_ptr(), _p(rscale, torch.float32), eff, M, N, ldc, alpha, 1 if accumulate else 0)
            deferred = (key, (rd, (c, rscale)))
        else:
            d.workspace = _p(_workspace(split_k * M * N, c.device), torch.float32)
    if deferred is not None:
        _PENDING.setdefault(deferred[0], []).append(deferred[1])
    if DEBUG_WS is not None and split_k <= 1:
        d.workspace = _p(DEBUG_WS, torch.float32)
        d.flags |= 2048 | DEBUG_FLAGS
    prof = PROFILE
    if prof is None:
        _lib.check(_lib.lib().toist_gemm_bf16(ctypes.byref(d), _stream()), "toist_gemm_bf16")
        return
    if not flops:
        flops = 2 * M * N * K * max(batch, 1)
    t = int(_lib.lib().toist_gemm_pick_tile(ctypes.byref(d)))
    key = (t, a_kind, b_kind)
    want = prof["key"]
    if want is not None and not (want(key) if callable(want) else (key == want or (isinstance(want, (set, frozenset)) and key in want))):
        _lib.check(_lib.lib().toist_gemm_bf16(ctypes.byref(d), _stream()), "toist_gemm_bf16")
        prof["other"][key] = prof["other"].get(key, 0) + flops
        return
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    _lib.check(_lib.lib().toist_gemm_bf16(ctypes.byref(d), _stream()), "toist_gemm_bf16")
    e1.record()
    ta = max(int(a.R) * int(a.S), 1) if a_kind in (A_CONV, A_CONVT) else 1
    tb = max(int(b.R) * int(b.S), 1) if b_kind == B_CONVX else 1
    nb = max(batch, 1)
    nbytes = 2 * M * K * nb // ta + 2 * N * K * nb // tb + (4 if c.dtype == torch.float32 else 2) * M * N * nb
    nbytes += 2 * M * N * nb * ((res is not None) + (aux is not None) + (pre_out is not None))
    prof["records"].append((e0, e1, flops, key, (M, N, K, nb, split_k, ta * tb), nbytes))


def matcher(logits, boxes, tgt_boxes, pos_map, tgt_off, match_off, max_T, w_class, w_bbox, w_giou, src_idx, tgt_idx,
            status, cost_out=None):
    L, B, Q, K = logits.shape
    _lib.check(
        _lib.lib().toist_matcher(_p(logits, torch.float32), _p(boxes, torch.float32), _p(tgt_boxes, torch.float32),
                                 _p(pos_map, torch.float32), _p(tgt_off, torch.int32), _p(match_off, torch.int32), L, B, Q,
                                 K, max_T, w_class, w_bbox, w_giou, _p(src_idx, torch.int64), _p(tgt_idx, torch.int64),
                                 _p(status, torch.int32), _p(cost_out, torch.float32), _stream()), "toist_matcher")


def layernorm_fwd(x, gamma, beta, eps, y, mean=None, rstd=None, add=None, y2=None):
    rows, D = x.shape
    _lib.check(
        _lib.lib().toist_layernorm_fwd(_p(x, torch.bfloat16), _p(gamma, torch.float32), _p(beta, torch.float32), eps, rows, D,
                                       _p(y, torch.bfloat16), _p(mean, torch.float32), _p(rstd, torch.float32), _p(add, torch.bfloat16),
                                       _p(y2, torch.bfloat16), _stream()),
        "toist_layernorm_fwd")


LN_DEFER = knob("TOIST_LN_DEFER", True)


def layernorm_bwd(dy, x, mean, rstd, gamma, dx, dgamma=None, dbeta=None, dx_drop=None, drop_p=0.0, seed=0, defer=False):
    """defer=True: dgamma / dbeta (f32, accumulated) are not updated by contended atomics inside the kernel -- every block writes its
    partial sums to the split-K arena and the fold is queued for flush_reductions() (engine.Tape.backward ends with it)."""
    rows, D = x.shape
    partials, blocks = None, 0
    if defer and LN_DEFER and dgamma is not None:
        key = (x.device, _raw_stream())
        if any(it[0].out in (dgamma.data_ptr(), dbeta.data_ptr()) for it in _PENDING.get(key, ())):
            flush_reductions()          # two queued folds into one output would race
        blocks = int(_lib.lib().toist_layernorm_bwd_blocks(rows))
        partials = _arena_take(2 * blocks * D, x.device)
        for i, out in enumerate((dgamma, dbeta)):
            rd = _lib.ReduceDesc(partials.data_ptr() + 4 * i * blocks * D, out.data_ptr(), None, blocks, 1, D, D, 1.0, 1)
            _PENDING.setdefault(key, []).append((rd, (out, partials)))
    _lib.check(
        _lib.lib().toist_layernorm_bwd(_p(dy, torch.bfloat16), _p(x, torch.bfloat16), _p(mean, torch.float32),
                                       _p(rstd, torch.float32), _p(gamma, torch.float32), rows, D, _p(dx, torch.bfloat16),
                                       None if partials is not None else _p(dgamma, torch.float32),
                                       None if partials is not None else _p(dbeta, torch.float32), _p(dx_drop, torch.bfloat16),
                                       drop_p, seed, _p(SEED_DEV), _p(partials, torch.float32), blocks, _stream()), "toist_layernorm_bwd")


def softmax_fwd(scores, key_pad, nbatch, H, Sq, Sk, ld, p, p_drop=None, drop_p=0.0, seed=0):
    _lib.check(
        _lib.lib().toist_softmax_fwd(_p(scores, torch.bfloat16), _p(key_pad, torch.uint8), nbatch, H, Sq, Sk, ld,
                                     _p(p, torch.bfloat16), _p(p_drop, torch.bfloat16), drop_p, seed, _p(SEED_DEV), _stream()),
        "toist_softmax_fwd")


def softmax_bwd(p, dp, rows, Sk, ld, ds, drop_p=0.0, seed=0):
    _lib.check(
        _lib.lib().toist_softmax_bwd(_p(p, torch.bfloat16), _p(dp, torch.bfloat16), rows, Sk, ld, _p(ds, torch.bfloat16),
                                     drop_p, seed, _p(SEED_DEV), _stream()), "toist_softmax_bwd")


def colsum(g, M, N, ld, out):
    _lib.check(_lib.lib().toist_colsum(_p(g, torch.bfloat16), M, N, ld, _p(out, torch.float32), _stream()), "toist_colsum")


def add(a, b, out, b_period=None):
    n = a.numel()
    _lib.check(
        _lib.lib().toist_add_bf16(_p(a, torch.bfloat16), _p(b, torch.bfloat16), n, n if b_period is None else b_period,
                                  _p(out, torch.bfloat16), _stream()), "toist_add_bf16")


STAMPS = None      # diagnostic (bench.py --stamps): {"buf": int64 device tensor, "names": [...]}: stamp(name) dates a point of the step on its stream


def stamp(name):
    st = STAMPS
    if st is None:
        return
    names = st["names"]
    if name in names:
        idx = names.index(name)
    else:
        idx = len(names)
        if idx >= st["buf"].numel():
            return
        names.append(name)
    _lib.check(_lib.lib().toist_stamp(_p(st["buf"], torch.int64), idx, _stream()), "toist_stamp")


def dropout(x, p, seed, out):
    _lib.check(_lib.lib().toist_dropout_bf16(_p(x, torch.bfloat16), x.numel(), p, seed, _p(SEED_DEV), _p(out, torch.bfloat16), _stream()),
               "toist_dropout_bf16")


def pack_image(nchw, out):
    N, C, H, W = nchw.shape
    _lib.check(_lib.lib().toist_pack_image(_p(nchw, torch.float32), N, C, H, W, _p(out, torch.bfloat16), _stream()), "toist_pack_image")


def stem_fwd(images, w, shift, out):
    """csrc/stem.hip: f32 NCHW images -> conv 7x7/2 (w bf16 [64,7,7,8], BN scale folded) + shift + ReLU + max-pool 3x3/2 -> bf16 NHWC `out`."""
    N, C, H, W = images.shape
    assert tuple(w.shape) == (64, 7, 7, 8) and w.is_contiguous() and images.is_contiguous() and out.is_contiguous()
    _lib.check(_lib.lib().toist_stem_fwd(_p(images, torch.float32), _p(w, torch.bfloat16), _p(shift, torch.float32), N, C, H, W,
                                         _p(out, torch.bfloat16), _stream()), "toist_stem_fwd")


def maxpool3x3s2(x, out):
    N, H, W, C = x.shape
    _lib.check(_lib.lib().toist_maxpool3x3s2(_p(x, torch.bfloat16), N, H, W, C, _p(out, torch.bfloat16), _stream()), "toist_maxpool3x3s2")


def unpack_nhwc(x, out):
    """bf16 [N, HW, C] -> f32 [N, C, HW]"""
    N, HW, C = x.shape
    _lib.check(_lib.lib().toist_unpack_nhwc(_p(x, torch.bfloat16), N, HW, C, _p(out, torch.float32), _stream()), "toist_unpack_nhwc")


def sine_position(mask_u8, num_pos_feats, temperature, out_tok=None, out_nchw=None):
    B, H, W = mask_u8.shape
    _lib.check(_lib.lib().toist_sine_position(_p(mask_u8, torch.uint8), B, H, W, num_pos_feats, temperature, _p(out_tok, torch.bfloat16),
                                              _p(out_nchw, torch.float32), _stream()), "toist_sine_position")


def text_prep(ids, attention_mask, pad_id):
    """-> (position ids int64 [B, L], key_pad uint8 [B, L]) of a tokenized batch"""
    B, L = ids.shape
    pos_ids = torch.empty(B, L, dtype=torch.int64, device=ids.device)
    key_pad = torch.empty(B, L, dtype=torch.uint8, device=ids.device)
    _lib.check(_lib.lib().toist_text_prep(_p(ids, torch.int64), _p(attention_mask, torch.int64), B, L, pad_id, _p(pos_ids), _p(key_pad), _stream()),
               "toist_text_prep")
    return pos_ids, key_pad


def sine_position_seq(mask_u8, num_pos_feats, temperature, out_tok):
    """out_tok bf16 [B, S, 2F], S >= H*W: rows H*W .. S-1 of every image are zero"""
    B, H, W = mask_u8.shape
    _lib.check(_lib.lib().toist_sine_position_seq(_p(mask_u8, torch.uint8), B, H, W, num_pos_feats, temperature, _p(out_tok, torch.bfloat16),
                                                  out_tok.shape[1], _stream()), "toist_sine_position_seq")


def embed_fwd(ids, pos_ids, word, pos, type0, out):
    n, D = out.shape
    _lib.check(_lib.lib().toist_embed_fwd(_p(ids, torch.int64), _p(pos_ids, torch.int64), _p(word, torch.float32), _p(pos, torch.float32),
                                          _p(type0, torch.float32), n, D, _p(out, torch.bfloat16), _stream()), "toist_embed_fwd")


def embed_bwd(g, ids, pos_ids, pad_id, dword, dpos, dtype0):
    n, D = g.shape
    _lib.check(_lib.lib().toist_embed_bwd(_p(g, torch.bfloat16), _p(ids, torch.int64), _p(pos_ids, torch.int64), n, D, pad_id,
                                          _p(dword, torch.float32), _p(dpos, torch.float32), _p(dtype0, torch.float32), _stream()),
               "toist_embed_bwd")


def criterion_fwd(logits, boxes, tgt_boxes, pos_map, tgt_off, match_off, src_idx, tgt_idx, num_boxes, eos_coef, losses, status=None):
    L, B, Q, K = logits.shape
    _lib.check(_lib.lib().toist_criterion_fwd(_p(logits, torch.float32), _p(boxes, torch.float32), _p(tgt_boxes, torch.float32),
                                              _p(pos_map, torch.float32), _p(tgt_off, torch.int32), _p(match_off, torch.int32),
                                              _p(src_idx, torch.int64), _p(tgt_idx, torch.int64), _p(num_boxes, torch.float32), L, B, Q, K,
                                              eos_coef, _p(losses, torch.float32), _p(status, torch.int32), _stream()), "toist_criterion_fwd")


def criterion_bwd(logits, boxes, tgt_boxes, pos_map, tgt_off, match_off, src_idx, tgt_idx, num_boxes, eos_coef, upstream, dlogits, dboxes):
    L, B, Q, K = logits.shape
    _lib.check(_lib.lib().toist_criterion_bwd(_p(logits, torch.float32), _p(boxes, torch.float32), _p(tgt_boxes, torch.float32),
                                              _p(pos_map, torch.float32), _p(tgt_off, torch.int32), _p(match_off, torch.int32),
                                              _p(src_idx, torch.int64), _p(tgt_idx, torch.int64), _p(num_boxes, torch.float32), L, B, Q, K,
                                              eos_coef, _p(upstream, torch.float32), _p(dlogits, torch.float32), _p(dboxes, torch.float32),
                                              _stream()), "toist_criterion_bwd")


def contrastive_fwd(pq, pt, tok_mask, tgt_off, match_off, src_idx, tgt_idx, num_boxes, temperature, losses):
    L, B, Q, D = pq.shape
    _lib.check(_lib.lib().toist_contrastive_fwd(_p(pq, torch.float32), _p(pt, torch.float32), _p(tok_mask, torch.int64), _p(tgt_off, torch.int32),
                                                _p(match_off, torch.int32), _p(src_idx, torch.int64), _p(tgt_idx, torch.int64),
                                                _p(num_boxes, torch.float32), L, B, Q, pt.shape[1], D, temperature, _p(losses, torch.float32),
                                                _stream()), "toist_contrastive_fwd")


def contrastive_bwd(pq, pt, tok_mask, tgt_off, match_off, src_idx, tgt_idx, num_boxes, temperature, upstream, dpq, dpt):
    L, B, Q, D = pq.shape
    _lib.check(_lib.lib().toist_contrastive_bwd(_p(pq, torch.float32), _p(pt, torch.float32), _p(tok_mask, torch.int64), _p(tgt_off, torch.int32),
                                                _p(match_off, torch.int32), _p(src_idx, torch.int64), _p(tgt_idx, torch.int64),
                                                _p(num_boxes, torch.float32), L, B, Q, pt.shape[1], D, temperature, _p(upstream, torch.float32),
                                                _p(dpq, torch.float32), _p(dpt, torch.float32), _stream()), "toist_contrastive_bwd")


def l2norm_fwd(x, y):
    rows, D = x.numel() // x.shape[-1], x.shape[-1]
    _lib.check(_lib.lib().toist_l2norm_fwd(_p(x, torch.float32), rows, D, _p(y, torch.float32), _stream()), "toist_l2norm_fwd")


def l2norm_bwd(x, dy, dx):
    rows, D = x.numel() // x.shape[-1], x.shape[-1]
    _lib.check(_lib.lib().toist_l2norm_bwd(_p(x, torch.float32), _p(dy, torch.float32), rows, D, _p(dx, torch.float32), _stream()), "toist_l2norm_bwd")


# ---- segmentation branch ---------------------------------------------------------------------------------
BF16, F32 = torch.bfloat16, torch.float32


def attnmap_softmax_fwd(scores, key_pad, B, Q, H, HW, ld, out):
    _lib.check(_lib.lib().toist_attnmap_softmax_fwd(_p(scores, BF16), _p(key_pad, torch.uint8), B, Q, H, HW, ld, _p(out, BF16), _stream()),
               "toist_attnmap_softmax_fwd")


def attnmap_softmax_bwd(prob, dprob, BQ, H, HW, ld, dscores):
    _lib.check(_lib.lib().toist_attnmap_softmax_bwd(_p(prob, BF16), _p(dprob, BF16), BQ, H, HW, ld, _p(dscores, BF16), _stream()),
               "toist_attnmap_softmax_bwd")


def groupnorm_fwd(x, gamma, beta, N, HW, C, G, eps, relu, y, stats):
    _lib.check(_lib.lib().toist_groupnorm_fwd(_p(x, BF16), _p(gamma, F32), _p(beta, F32), N, HW, C, G, eps, 1 if relu else 0, _p(y, BF16),
                                              _p(stats, F32), _stream()), "toist_groupnorm_fwd")


def groupnorm_apply(x, stats, gamma, beta, N, HW, C, G, eps, relu, y):
    _lib.check(_lib.lib().toist_groupnorm_apply(_p(x, BF16), _p(stats, F32), _p(gamma, F32), _p(beta, F32), N, HW, C, G, eps, 1 if relu else 0, _p(y, BF16),
                                                _stream()), "toist_groupnorm_apply")


def groupnorm_bwd(dy, y, x, stats, gamma, N, HW, C, G, eps, relu, dx, dgamma, dbeta, bstats, beta=None):
    """y = None with beta given: the ReLU mask is re-derived from x (two passes over the activation fewer)"""
    _lib.check(_lib.lib().toist_groupnorm_bwd(_p(dy, BF16), _p(y, BF16), _p(x, BF16), _p(stats, F32), _p(gamma, F32), _p(beta, F32), N, HW, C, G, eps,
                                              1 if relu else 0, _p(dx, BF16), _p(dgamma, F32), _p(dbeta, F32), _p(bstats, F32), _stream()),
               "toist_groupnorm_bwd")


def upsample_add(inp, fpn, BQ, Q, H, W, C, out):
    _lib.check(_lib.lib().toist_upsample_add(_p(inp, BF16), _p(fpn, BF16), BQ, Q, H, W, C, _p(out, BF16), _stream()), "toist_upsample_add")


def resize_add(inp, fpn, rows, n, Q, H, W, OH, OW, C, out):
    """out [n, OH, OW, C] = fpn[image] + nearest-resized inp [n, H, W, C] (any target size; rows = None or the int64 map indices of a gathered subset)."""
    _lib.check(_lib.lib().toist_resize_add(_p(inp, BF16), _p(fpn, BF16), _p(rows, torch.int64), n, Q, H, W, OH, OW, C, _p(out, BF16), _stream()), "toist_resize_add")


def resize_add_bwd(dout, BQ, H, W, OH, OW, C, din):
    _lib.check(_lib.lib().toist_resize_add_bwd(_p(dout, BF16), BQ, H, W, OH, OW, C, _p(din, BF16), _stream()), "toist_resize_add_bwd")


def upsample_add_rows(inp, fpn, rows, n, Q, H, W, C, out):
    """upsample_add on gathered maps: map i of inp / out is map rows[i] (int64, device) of the batch."""
    _lib.check(_lib.lib().toist_upsample_add_rows(_p(inp, BF16), _p(fpn, BF16), _p(rows, torch.int64), n, Q, H, W, C, _p(out, BF16), _stream()),
               "toist_upsample_add_rows")


def mask_stage_fwd(src, src_stats, gamma, beta, fpn_conv, w, bias, out, out_stats, N, Q, H, W, c_in, c_out, w_rows, gn_in, up, eps=1e-5):
    """One fused stage of the mask head's tail (csrc/maskstage.hip): [GroupNorm + ReLU of the source] -> [2x upsample] -> 3x3 conv [+ fpn_conv, the
    image's lay(adapter(fpn)) [N/Q,H,W,c_out]] -> raw output + its GroupNorm sums (c_out > 1: out bf16 [N,H,W,c_out]) or the f32 mask logits
    (c_out == 1: out f32 [N,H,W])."""
    _lib.check(_lib.lib().toist_mask_stage_fwd(_p(src, BF16), _p(src_stats, F32), _p(gamma, F32), _p(beta, F32), _p(fpn_conv, BF16), _p(w, BF16), _p(bias, F32),
                                               _p(out, F32 if c_out == 1 else BF16), _p(out_stats, F32), N, Q, H, W, c_in, c_out, w_rows,
                                               1 if gn_in else 0, 1 if up else 0, eps, _stream()), "toist_mask_stage_fwd")


def upsample_add_bwd(dout, BQ, H, W, C, din):
    _lib.check(_lib.lib().toist_upsample_add_bwd(_p(dout, BF16), BQ, H, W, C, _p(din, BF16), _stream()), "toist_upsample_add_bwd")


def sum_queries(inp, B, Q, per, out):
    _lib.check(_lib.lib().toist_sum_queries(_p(inp, BF16), B, Q, per, _p(out, BF16), _stream()), "toist_sum_queries")


def mask_loss_fwd(pred, pred_row, gt, gt_row, T, h, w, TH, TW, alpha, sums, valid_hw=None):
    """valid_hw: optional device int32 [4] = {VH, VW, hs, ws}, the batch's own corner of a larger bucket (include/toist_hip.h)."""
    _lib.check(_lib.lib().toist_mask_loss_fwd(_p(pred, F32), _p(pred_row, torch.int32), _p(gt, torch.uint8), _p(gt_row, torch.int32), T, h, w, TH,
                                              TW, alpha, _p(sums, F32), _p(valid_hw, torch.int32), _stream()), "toist_mask_loss_fwd")


def sum_segments(inp, seg, B, rows, per, out):
    """out[b] = sum of the maps inp[seg[b]:seg[b+1]] (seg int32 [B+1] on the device)."""
    _lib.check(_lib.lib().toist_sum_segments(_p(inp, BF16), _p(seg, torch.int32), B, rows, per, _p(out, BF16), _stream()), "toist_sum_segments")


def mask_loss_bwd(pred, pred_row, gt, gt_row, T, h, w, TH, TW, alpha, sums, coef, dpred, compact=False, valid_hw=None):
    if compact:     # dpred is [T,h,w]: pair t's gradient in row t
        _lib.check(_lib.lib().toist_mask_loss_bwd_compact(_p(pred, F32), _p(pred_row, torch.int32), _p(gt, torch.uint8), _p(gt_row, torch.int32), T, h, w,
                                                          TH, TW, alpha, _p(sums, F32), _p(coef, F32), _p(dpred, F32), _p(valid_hw, torch.int32), _stream()), "toist_mask_loss_bwd_compact")
        return
    _lib.check(_lib.lib().toist_mask_loss_bwd(_p(pred, F32), _p(pred_row, torch.int32), _p(gt, torch.uint8), _p(gt_row, torch.int32), T, h, w, TH,
                                              TW, alpha, _p(sums, F32), _p(coef, F32), _p(dpred, F32), _p(valid_hw, torch.int32), _stream()), "toist_mask_loss_bwd")


# ---- optimizer tail (include/toist_hip.h: toist_opt_*) ---------------------------------------------------------
def opt_chunk_elems():
    return int(_lib.lib().toist_opt_chunk_elems())


def opt_sqnorm(table, grads, chunks, n_chunks, partial):
    _lib.check(_lib.lib().toist_opt_sqnorm(_p(table, torch.uint8), _p(grads, torch.int64), _p(chunks, torch.int32), n_chunks,
                                           _p(partial, torch.float32), _stream()), "toist_opt_sqnorm")


def opt_finish_norm(partial, n_chunks, max_norm, beta1, beta2, state):
    _lib.check(_lib.lib().toist_opt_finish_norm(_p(partial, torch.float32), n_chunks, max_norm, beta1, beta2, _p(state, torch.uint8),
                                                _stream()), "toist_opt_finish_norm")


def opt_adamw_ema(table, grads, chunks, n_chunks, groups, state, beta1, beta2, eps, ema_decay, max_blocks=0):
    _lib.check(_lib.lib().toist_opt_adamw_ema_blocks(_p(table, torch.uint8), _p(grads, torch.int64), _p(chunks, torch.int32), n_chunks,
                                                     _p(groups, torch.float32), _p(state, torch.uint8), beta1, beta2, eps, ema_decay, max_blocks, _stream()),
               "toist_opt_adamw_ema")


def lsap(cost, offset, rows, cols, n, max_rows, max_cols, max_cells, out_off, row_idx, col_idx, status, ld=0):
    _lib.check(_lib.lib().toist_lsap(_p(cost, torch.float32), _p(offset, torch.int64), _p(rows, torch.int32), _p(cols, torch.int32), ld, n, max_rows,
                                     max_cols, max_cells, _p(out_off, torch.int64), _p(row_idx, torch.int64), _p(col_idx, torch.int64),
                                     _p(status, torch.int32), _stream()), "toist_lsap")


def conv3x3_small(dgrad, src, w, shift, res, out, n_img, H, W, c_src, c_out):
    """3x3 / s1 / p1 convolution (or its data gradient) with <= 32 channels on both sides; w is [Co,3,3,Ci] bf16."""
    w_co, w_ci = int(w.shape[0]), int(w.shape[3])
    _lib.check(_lib.lib().toist_conv3x3_small(1 if dgrad else 0, _p(src, torch.bfloat16), _p(w, torch.bfloat16), _p(shift, torch.float32),
                                              _p(res, torch.bfloat16), _p(out, torch.bfloat16), n_img, H, W, c_src, c_out, w_co, w_ci, _stream()),
               "toist_conv3x3_small")


def wgrad3x3_small(dy, x, out, defer=False, accumulate=True):
    """out [Co,3,3,C] (f32) += weight gradient of a 3x3 / s1 / p1 convolution with C in {16, 32}, Co in {8, 16} (csrc/smallconv.hip):
    per-workgroup partials into the split-K arena, folded by the batched reduction (queued when defer=True)."""
    n_img, H, W, C = x.shape
    Co = dy.shape[-1]
    blocks = int(_lib.lib().toist_wgrad3x3_small_blocks())
    ws = _arena_take(blocks * Co * 9 * C, x.device)
    _lib.check(_lib.lib().toist_wgrad3x3_small(_p(dy, torch.bfloat16), _p(x, torch.bfloat16), _p(ws, torch.float32), n_img, H, W, C, Co, _stream()),
               "toist_wgrad3x3_small")
    key = (x.device, _raw_stream())
    if any(it[0].out == out.data_ptr() for it in _PENDING.get(key, ())):
        flush_reductions()
    rd = _lib.ReduceDesc(ws.data_ptr(), out.data_ptr(), None, blocks, Co, 9 * C, 9 * C, 1.0, 1 if accumulate else 0)
    _PENDING.setdefault(key, []).append((rd, (out,)))
    if not defer:
        flush_reductions()


def attn2_splits(Sk):
    return int(_lib.lib().toist_attn2_splits(Sk))


def attn2_fwd(q, kmat, v, key_pad, B, H, Sq, Sk, dh, scale, drop_p, seed, ctx, lse):
    """Flash-style attention core (csrc/attn2.hip): q / k / v / ctx are column slices of packed [B*S, ld] bf16 buffers; lse f32 [B*H, Sq, 2]."""
    _lib.check(_lib.lib().toist_attn2_fwd(_p(q, torch.bfloat16), q.stride(0), _p(kmat, torch.bfloat16), kmat.stride(0), _p(v, torch.bfloat16), v.stride(0),
                                          _p(key_pad, torch.uint8), B, H, Sq, Sk, dh, scale, drop_p, seed, _p(SEED_DEV) if drop_p > 0 else None,
                                          _p(ctx, torch.bfloat16), ctx.stride(0), _p(lse, torch.float32), _stream()), "toist_attn2_fwd")


def attn2_bwd(q, kmat, v, ctx, dctx, lse, key_pad, B, H, Sq, Sk, dh, scale, drop_p, seed, dq, dk, dv, dq_part=None):
    """Key-owning backward of attn2_fwd: dk / dv written once; dq directly when attn2_splits(Sk) == 1, else its per-split shares go to
    dq_part (bf16 [splits, B*Sq, H*dh]) for the consumer's fold (kernels.rowgemm(fold=...))."""
    _lib.check(_lib.lib().toist_attn2_bwd(_p(q, torch.bfloat16), q.stride(0), _p(kmat, torch.bfloat16), kmat.stride(0), _p(v, torch.bfloat16), v.stride(0),
                                          _p(ctx, torch.bfloat16), ctx.stride(0), _p(dctx, torch.bfloat16), dctx.stride(0), _p(lse, torch.float32),
                                          _p(key_pad, torch.uint8), B, H, Sq, Sk, dh, scale, drop_p, seed, _p(SEED_DEV) if drop_p > 0 else None,
                                          _p(dq, torch.bfloat16), dq.stride(0) if dq is not None else 0, _p(dk, torch.bfloat16), dk.stride(0),
                                          _p(dv, torch.bfloat16), dv.stride(0), _p(dq_part, torch.bfloat16), _stream()), "toist_attn2_bwd")


def rowgemm(a, w, out, *, b_kind, epi, K=None, bias=None, res=None, res2=None, drop_p=0.0, drop_seed=0, gamma=None, beta=None, eps=1e-5,
            z=None, mean=None, rstd=None, add=None, out2=None, fold=None, fold_cols=0, dgamma=None, dbeta=None):
    """Row-complete sub-layer launch (csrc/tlayer.hip, include/toist_hip.h: toist_rowgemm): out[M, 256] from a[M, K] and the nn.Linear
    weight copy `w` read in place (b_kind B_ROWK = forward, B_KROW = data gradient), finished on chip by epi = ROW_PLAIN / ROW_LN_FWD /
    ROW_LN_BWD.  fold = bf16 [parts, M, fold_cols] partial sums that replace (and are written back into) a[:, :fold_cols].
    ROW_LN_BWD: dgamma / dbeta (f32 [256], accumulated) receive the parameter gradients through per-block partials queued for
    flush_reductions(), exactly like layernorm_bwd(defer=True)."""
    M = a.shape[0]
    d = _lib.RowGemm()
    d.M, d.K, d.b_kind, d.epi = M, (a.shape[1] if K is None else K), b_kind, epi
    d.a, d.lda, d.w, d.ldw = _p(a, torch.bfloat16), a.stride(0), _p(w, torch.bfloat16), w.stride(0)
    if fold is not None:
        d.fold, d.fold_parts, d.fold_cols, d.fold_stride = _p(fold, torch.bfloat16), fold.shape[0], fold_cols, fold.stride(0)
        assert fold.shape[1] == M and fold.shape[2] == fold_cols and fold.stride(1) == fold_cols
    d.bias = _p(bias, torch.float32)
    if res is not None:
        d.res, d.ldr = _p(res, torch.bfloat16), res.stride(0)
    if res2 is not None:
        d.res2, d.ldr2 = _p(res2, torch.bfloat16), res2.stride(0)
    d.drop_p, d.eps, d.drop_seed = drop_p, eps, drop_seed
    d.drop_seed_dev = _p(SEED_DEV) if drop_p > 0 else None
    d.gamma, d.beta = _p(gamma, torch.float32), _p(beta, torch.float32)
    for name, t in (("z", z), ("add", add), ("out2", out2)):
        if t is not None:
            assert t.is_contiguous() and t.shape[-1] == 256
            setattr(d, name, _p(t, torch.bfloat16))
    d.mean, d.rstd = _p(mean, torch.float32), _p(rstd, torch.float32)
    d.out, d.ldo = _p(out, torch.bfloat16), out.stride(0)
    keep = None
    if epi == _lib.ROW_LN_BWD and dgamma is not None:
        key = (a.device, _raw_stream())
        if any(it[0].out in (dgamma.data_ptr(), dbeta.data_ptr()) for it in _PENDING.get(key, ())):
            flush_reductions()          # two queued folds into one output would race
        blocks = int(_lib.lib().toist_rowgemm_blocks(M))
        partials = _arena_take(2 * blocks * 256, a.device)
        for i, o in enumerate((dgamma, dbeta)):
            rd = _lib.ReduceDesc(partials.data_ptr() + 4 * i * blocks * 256, o.data_ptr(), None, blocks, 1, 256, 256, 1.0, 1)
            _PENDING.setdefault(key, []).append((rd, (o, partials)))
        d.partials = _p(partials, torch.float32)
        keep = partials
    _lib.check(_lib.lib().toist_rowgemm(ctypes.byref(d), _stream()), "toist_rowgemm")
    return keep


ROW_PLAIN, ROW_LN_FWD, ROW_LN_BWD = _lib.ROW_PLAIN, _lib.ROW_LN_FWD, _lib.ROW_LN_BWD


# ---- XCD-resident decoder stack (csrc/xdec.hip) ------------------------------------------------------------------------------------------
XDEC_PROF = None
XDEC_TEST_ABSENT = 0    # failure-path tests only (tests/test_gpu_xdec.py): that many workgroups per XCD leave the launch at once
XDEC_LAUNCHES = 0       # forward launches issued (or captured) so far: Transformer.decode_tokens learns from it whether its program used one
XDEC_FAILED = False     # process-wide: a launch reported an expired spin (xdec_check) -> the decoder runs on the per-op launches from then on
_XDEC_CTL = {}          # device -> int32[1024] control words (tickets, arrival counters, sticky status)


def _xdec_ctl(device, create=True):
    """One set of control words per device: the launches need the whole chip, so two of them never run at the same time on one device (the forward and
    the backward launch of a step are ordered by the stream).  Allocated EAGERLY: words taken from a capturing graph's pool would be re-zeroed by every
    replay, sticky status word included -- xdec_supported() therefore refuses the launches while a capture is running and no words exist yet
    (a capture is always preceded by an eager warm-up step in toist_amd.harness / bench.py, which allocates them)."""
    ctl = _XDEC_CTL.get(device)
    if ctl is None and create:
        if torch.cuda.is_current_stream_capturing():
            return None
        ctl = _XDEC_CTL[device] = torch.zeros(_lib.XDEC_CTL_WORDS, dtype=torch.int32, device=device)
    return ctl


def _ranks_share_a_device():
    """True when the ranks of this job on this node cannot have one GPU each (the XCD-resident launches need all 256 CUs to themselves).  LOCAL_WORLD_SIZE is
    what torchrun / the driver's launch line export; without it (a hand-made rendezvous) the world size is the best bound there is."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return False
    local = os.environ.get("LOCAL_WORLD_SIZE")
    local = int(local) if local and local.isdigit() else dist.get_world_size()
    return local > torch.cuda.device_count()


def xdec_supported(B, Q, S, L, ff=2048):
    """The XCD-resident decoder launches need all 256 CUs of the device to themselves (one 147 KB workgroup per CU, 32 co-resident per image): not
    when several ranks of one job share a GPU (the one-GPU multi-rank tests; two such launches from two processes would starve each other until their
    bounded spins expire), not after a launch has reported an expired spin (XDEC_FAILED), not for a dim_feedforward other than the compiled-in 2048
    (ADVICE r5), and never with TOIST_XDEC=0 (the opt-out for a GPU shared with another process or a CU-masked queue: INTEGRATION.md)."""
    if XDEC_FAILED or ff != 2048 or _ranks_share_a_device():
        return False
    if not _lib.lib().toist_xdec_supported(B, Q, S, L):
        return False
    return _xdec_ctl(torch.device("cuda", torch.cuda.current_device())) is not None


def xdec_fwd(B, Q, S, x0, qpos, kv, key_pad, drop_p, eps, out, layers, part):
    """All decoder layers in one launch (include/toist_hip.h: toist_xdec_fwd).  `out` = dict of the stacked per-layer tensors named as the
    descriptor's fields; `layers` = one dict per layer with the 18 parameter tensors + "seed" (6 ints)."""
    d = _lib.Xdec()
    d.B, d.Q, d.S, d.L = B, Q, S, len(layers)
    d.ff, d.test_absent = int(layers[0]["w1"].shape[0]), XDEC_TEST_ABSENT
    assert all(tuple(ly["w1"].shape) == (d.ff, 256) and tuple(ly["w2"].shape) == (256, d.ff) for ly in layers)
    d.x0, d.qpos, d.kv, d.ldkv = _p(x0, torch.bfloat16), _p(qpos, torch.bfloat16), _p(kv, torch.bfloat16), kv.stride(0)
    d.key_pad = _p(key_pad, torch.uint8) if key_pad is not None else None
    d.drop_p, d.eps = drop_p, eps
    d.seed_dev = _p(SEED_DEV) if drop_p > 0 else None
    for name in ("qkv", "ctx_s", "z1", "y1", "y1e", "qc", "ctx_c", "z3", "y3", "h", "z4", "y4", "y4e"):
        t = out[name]
        assert t.is_contiguous()
        setattr(d, name, _p(t, torch.bfloat16))
    for name in ("lse_s", "mean1", "rstd1", "lse_c", "mean3", "rstd3", "mean4", "rstd4"):
        t = out[name]
        assert t.is_contiguous()
        setattr(d, name, _p(t, torch.float32))
    assert part.is_contiguous() and part.numel() >= B * 32 * 128 * 256
    d.part = _p(part, torch.bfloat16)
    d.ctl = _xdec_ctl(x0.device).data_ptr()       # (never None here: xdec_supported() said yes)
    d.prof = XDEC_PROF.data_ptr() if XDEC_PROF is not None else None      # diagnostics (tools/r5/xdec_bench.py): int64 [256, L, 8]
    for i, ly in enumerate(layers):
        e = d.layer[i]
        for name in ("w_in", "w_os", "w_q", "w_oc", "w1", "w2"):
            t = ly[name]
            assert t.is_contiguous()
            setattr(e, name, _p(t, torch.bfloat16))
        for name in ("b_in", "b_os", "g1", "be1", "b_q", "b_oc", "g3", "be3", "b1", "b2", "g4", "be4"):
            t = ly[name]
            assert t.is_contiguous()
            setattr(e, name, _p(t, torch.float32))
        for j in range(6):
            e.seed[j] = ly["seed"][j]
    global XDEC_LAUNCHES
    XDEC_LAUNCHES += 1
    _lib.check(_lib.lib().toist_xdec_fwd(ctypes.byref(d), _stream()), "toist_xdec_fwd")


def xdec_bwd(B, Q, S, kv, key_pad, drop_p, saved, g_out, outs, layers, scratch):
    """Data-gradient chain of all decoder layers in one launch (include/toist_hip.h: toist_xdec_bwd).  saved = the forward's stacked tensors,
    g_out bf16 [L, B*Q, 256], outs = dict(gb4, dh, go3, go1, sink, dkv, ln_part), scratch = dict(dctx, part, dq_part)."""
    d = _lib.XdecBwd()
    d.B, d.Q, d.S, d.L = B, Q, S, len(layers)
    d.ff, d.test_absent = int(layers[0]["w1"].shape[0]), XDEC_TEST_ABSENT
    d.kv, d.ldkv = _p(kv, torch.bfloat16), kv.stride(0)
    d.key_pad = _p(key_pad, torch.uint8) if key_pad is not None else None
    d.drop_p = drop_p
    d.seed_dev = _p(SEED_DEV) if drop_p > 0 else None
    for name in ("qkv", "ctx_s", "z1", "qc", "ctx_c", "z3", "h", "z4"):
        assert saved[name].is_contiguous()
        setattr(d, name, _p(saved[name], torch.bfloat16))
    for name in ("lse_s", "mean1", "rstd1", "lse_c", "mean3", "rstd3", "mean4", "rstd4"):
        assert saved[name].is_contiguous()
        setattr(d, name, _p(saved[name], torch.float32))
    assert g_out.is_contiguous() and g_out.shape == (len(layers), B * Q, 256)
    d.g_out = _p(g_out, torch.bfloat16)
    for name in ("gb4", "dh", "go3", "go1"):
        assert outs[name].is_contiguous()
        setattr(d, name, _p(outs[name], torch.bfloat16))
    sink, dkv = outs["sink"], outs["dkv"]
    assert sink.stride(1) == 1 and dkv.stride(1) == 1
    d.sink, d.ldsink, d.dkv, d.lddkv = _p(sink, torch.bfloat16), sink.stride(0), _p(dkv, torch.bfloat16), dkv.stride(0)
    d.ln_part = _p(outs["ln_part"], torch.float32)
    for name in ("dctx", "part", "dq_part"):
        assert scratch[name].is_contiguous()
        setattr(d, name, _p(scratch[name], torch.bfloat16))
    d.ctl = _xdec_ctl(kv.device).data_ptr()
    d.prof = XDEC_PROF.data_ptr() if XDEC_PROF is not None else None
    for i, ly in enumerate(layers):
        e = d.layer[i]
        for name in ("w_in", "w_os", "w_q", "w_oc", "w1", "w2"):
            assert ly[name].is_contiguous()
            setattr(e, name, _p(ly[name], torch.bfloat16))
        for name in ("g1", "g3", "g4"):
            setattr(e, name, _p(ly[name], torch.float32))
        for j in range(6):
            e.seed[j] = ly["seed"][j]
    _lib.check(_lib.lib().toist_xdec_bwd(ctypes.byref(d), _stream()), "toist_xdec_bwd")


def queue_fold(partials, out, splits, keep=()):
    """out[256] += sum over `splits` rows of partials [splits, 256] (f32), folded with the other deferred reductions of the current stream
    (flush_reductions): the LayerNorm parameter gradients of the XCD-resident backward."""
    key = (out.device, _raw_stream())
    if any(it[0].out == out.data_ptr() for it in _PENDING.get(key, ())):
        flush_reductions()
    rd = _lib.ReduceDesc(partials.data_ptr(), out.data_ptr(), None, splits, 1, 256, 256, 1.0, 1)
    _PENDING.setdefault(key, []).append((rd, (out, partials) + tuple(keep)))


def xdec_check(raise_on_failure=True):
    """The sticky status words of the XCD-resident launches (synchronises: one 4-byte read per device).  A set word means a bounded spin expired -- the 32
    workgroups of an image were not co-resident (a second process on the GPU, a CU-masked queue, an unexpected dispatch order) -- and that launch's
    results are invalid (the launch also turned them into NaN, so the losses of that step are NaN).  The word is cleared, XDEC_FAILED is set so that
    every later decoder forward / backward runs on the per-op launches (graphs captured with the launches inside must be re-captured:
    harness.CapturedTrainStep does), and RuntimeError is raised unless raise_on_failure is False (then the return value says whether a launch failed).
    Called by harness.finite_or_exit, CapturedTrainStep when a replayed loss is not finite, harness.evaluate, Transformer's eval-mode decode, bench.py,
    smoke()."""
    global XDEC_FAILED
    failed = False
    for dev, ctl in _XDEC_CTL.items():
        if int(ctl[_lib.XDEC_CTL_WORDS - 1].item()) != 0:
            failed = True
            ctl[_lib.XDEC_CTL_WORDS - 1] = 0
    if failed:
        XDEC_FAILED = True
        if raise_on_failure:
            raise RuntimeError("toist_xdec: a workgroup gave up waiting for its XCD group (the 32 workgroups of an image were not co-resident); the results of "
                               "that step are invalid (NaN).  The XCD-resident decoder launches are now off for this process: repeat the step (per-op launches), "
                               "or start with TOIST_XDEC=0 when the GPU is shared")
    return failed


def scatter_rows(src, src_row, dst, dst_row):
    """dst[dst_row[i]] = src[src_row[i]] (f32 rows) for the i with both indices >= 0 (int64 device vectors); see toist_scatter_rows_f32."""
    assert src.is_contiguous() and dst.is_contiguous() and src.shape[-1] == dst.shape[-1]
    _lib.check(_lib.lib().toist_scatter_rows_f32(_p(src, torch.float32), _p(src_row, torch.int64), _p(dst, torch.float32), _p(dst_row, torch.int64), src_row.numel(),
                                                 src.shape[-1], _stream()), "toist_scatter_rows_f32")


def kmeans(banks, centers, group_task, group_off, members, features, tol, max_iter, pick, chosen_center, iters=None):
    """banks [T, N, D] f32, centers [T, K, D] f32 (updated in place); see toist_kmeans."""
    T, N, D = banks.shape
    K = centers.shape[1]
    _lib.check(_lib.lib().toist_kmeans(_p(banks, torch.float32), banks.stride(0), _p(centers, torch.float32), centers.stride(0), _p(group_task, torch.int32),
                                       _p(group_off, torch.int32), _p(members, torch.int32), group_task.numel(), _p(features, torch.float32), N, D, K, tol,
                                       max_iter, _p(pick, torch.int32), _p(chosen_center, torch.float32), _p(iters, torch.int32), _stream()), "toist_kmeans")


def attn_small_fwd(q, kmat, v, key_pad, B, H, S, dh, scale, drop_p, seed, ctx, stats, bq=None, bk=None, bv=None):
    """Whole-head self-attention for S <= 64, dh <= 64 (csrc/attn_small.hip); column slices of packed buffers, optional projection biases."""
    _lib.check(_lib.lib().toist_attn_small_fwd(_p(q, torch.bfloat16), q.stride(0), _p(kmat, torch.bfloat16), kmat.stride(0), _p(v, torch.bfloat16), v.stride(0),
                                               _p(key_pad, torch.uint8), B, H, S, dh, scale, drop_p, seed, _p(SEED_DEV) if drop_p > 0 else None,
                                               _p(ctx, torch.bfloat16), ctx.stride(0), _p(stats, torch.float32), _p(bq, torch.float32), _p(bk, torch.float32),
                                               _p(bv, torch.float32), _stream()), "toist_attn_small_fwd")


def attn_small_bwd(q, kmat, v, key_pad, B, H, S, dh, scale, drop_p, seed, stats, dctx, dq, dk, dv, bq=None, bk=None, bv=None):
    _lib.check(_lib.lib().toist_attn_small_bwd(_p(q, torch.bfloat16), q.stride(0), _p(kmat, torch.bfloat16), kmat.stride(0), _p(v, torch.bfloat16), v.stride(0),
                                               _p(key_pad, torch.uint8), B, H, S, dh, scale, drop_p, seed, _p(SEED_DEV) if drop_p > 0 else None,
                                               _p(stats, torch.float32), _p(dctx, torch.bfloat16), dctx.stride(0), _p(dq, torch.bfloat16), dq.stride(0),
                                               _p(dk, torch.bfloat16), dk.stride(0), _p(dv, torch.bfloat16), dv.stride(0), _p(bq, torch.float32),
                                               _p(bk, torch.float32), _p(bv, torch.float32), _stream()), "toist_attn_small_bwd")


# ---- evaluation masks (csrc/evalmask.hip): column-major bit planes [n, W, ceil(H/64)] stored in int64 tensors --------------
def mask_words(h):
    return (h + 63) // 64


def _plane(n, h, w, device):
    if h * w >= 2 ** 31:
        raise ValueError("mask planes address pixels with 31 bits")
    return torch.empty(n, w, mask_words(h), dtype=torch.int64, device=device)


def mask_resize_pack(logits, max_size, crop, out_size, threshold=0.5):
    """[n, h0, w0] fp32 mask logits -> bit planes of sigmoid(resize(resize(.)[:crop])) > threshold (PostProcessSegm's arithmetic)."""
    n, h0, w0 = logits.shape
    bits = _plane(n, out_size[0], out_size[1], logits.device)
    _lib.check(_lib.lib().toist_mask_resize_pack(_p(logits.contiguous(), torch.float32), n, h0, w0, int(max_size[0]), int(max_size[1]), int(crop[0]),
                                                 int(crop[1]), int(out_size[0]), int(out_size[1]), float(threshold), _p(bits), _stream()),
               "toist_mask_resize_pack")
    return bits


def mask_pack(dense):
    """[n, h, w] bool / uint8 -> bit planes."""
    n, h, w = dense.shape
    src = dense.contiguous().view(torch.uint8) if dense.dtype == torch.bool else dense.contiguous()
    bits = _plane(n, h, w, dense.device)
    _lib.check(_lib.lib().toist_mask_pack(_p(src, torch.uint8), n, h, w, _p(bits), _stream()), "toist_mask_pack")
    return bits


def mask_unpack(bits, h, w):
    n = bits.shape[0]
    dense = torch.empty(n, h, w, dtype=torch.uint8, device=bits.device)
    _lib.check(_lib.lib().toist_mask_unpack(_p(bits, torch.int64), n, h, w, _p(dense), _stream()), "toist_mask_unpack")
    return dense.view(torch.bool)


def mask_area(bits, h, w):
    area = torch.empty(bits.shape[0], dtype=torch.int32, device=bits.device)
    _lib.check(_lib.lib().toist_mask_area(_p(bits, torch.int64), bits.shape[0], h, w, _p(area), _stream()), "toist_mask_area")
    return area


def mask_iou(dt, gt, iscrowd, area_dt, area_gt, h, w):
    """[n_dt, n_gt] float64 IoU of two sets of bit planes (crowd ground truth: union = detection area)."""
    iou = torch.zeros(dt.shape[0], gt.shape[0], dtype=torch.float64, device=dt.device)
    _lib.check(_lib.lib().toist_mask_iou(_p(dt, torch.int64), dt.shape[0], _p(gt, torch.int64), gt.shape[0], _p(iscrowd, torch.uint8),
                                         _p(area_dt, torch.int32), _p(area_gt, torch.int32), h, w, _p(iou), _stream()), "toist_mask_iou")
    return iou


def mask_rle(bits, h, w):
    """Run lengths of every plane, zeros first (mask_util.encode's counts): (counts int32 [total], first_run int64 [n + 1])."""
    n = bits.shape[0]
    trans = torch.empty(n, w, dtype=torch.int32, device=bits.device)
    _lib.check(_lib.lib().toist_mask_rle_count(_p(bits, torch.int64), n, h, w, _p(trans), _stream()), "toist_mask_rle_count")
    ends = torch.cumsum(trans.view(-1).to(torch.int64), 0)
    col_off = ends - trans.view(-1)
    first_pos = torch.zeros(n + 1, dtype=torch.int64, device=bits.device)
    first_pos[1:] = ends.view(n, w)[:, -1]
    first_run = first_pos + torch.arange(n + 1, device=bits.device)
    total = int(first_pos[-1])                       # the one synchronisation: sizes the output
    pos = torch.empty(max(total, 1), dtype=torch.int32, device=bits.device)
    _lib.check(_lib.lib().toist_mask_rle_emit(_p(bits, torch.int64), n, h, w, _p(col_off), _p(pos), _stream()), "toist_mask_rle_emit")
    counts = torch.empty(total + n, dtype=torch.int32, device=bits.device)
    _lib.check(_lib.lib().toist_mask_rle_counts(_p(pos), _p(first_pos), _p(first_run), n, h, w, _p(counts), _stream()), "toist_mask_rle_counts")
    return counts, first_run


def coco_match(iou, iou_off, dt_area, dt_off, gt_area, gt_ignore, gt_crowd, gt_off, area_rng, thrs):
    """Batched COCOeval.evaluateImg (csrc/evalmask.hip): -> (dt_match int32, dt_ignore uint8) flat [sum_i A*T*D_i],
    gt_range_ignore uint8 flat [sum_i A*G_i]; offsets are int64 device tensors with n_images + 1 entries."""
    n_img, A, T = dt_off.numel() - 1, area_rng.shape[0], thrs.numel()
    n_dt, n_gt = dt_area.numel(), gt_area.numel()
    dev = dt_off.device
    dt_match = torch.empty(A * T * n_dt, dtype=torch.int32, device=dev)
    dt_ignore = torch.empty(A * T * n_dt, dtype=torch.uint8, device=dev)
    gt_flag = torch.empty(A * n_gt, dtype=torch.uint8, device=dev)
    taken = torch.empty(A * T * n_gt, dtype=torch.uint8, device=dev)
    _lib.check(_lib.lib().toist_coco_match(_p(iou, torch.float64), _p(iou_off, torch.int64), _p(dt_area, torch.float64), _p(dt_off, torch.int64),
                                           _p(gt_area, torch.float64), _p(gt_ignore, torch.uint8), _p(gt_crowd, torch.uint8), _p(gt_off, torch.int64),
                                           n_img, _p(area_rng, torch.float64), A, _p(thrs, torch.float64), T, _p(dt_match), _p(dt_ignore),
                                           _p(gt_flag), _p(taken), _stream()), "toist_coco_match")
    return dt_match, dt_ignore, gt_flag
