"""ctypes binding of libtoist_hip.so (include/toist_hip.h).

The product path has exactly one backend: the hand-written HIP library.  If it is missing or fails
to load, importing a kernel raises -- there is no CPU or eager-PyTorch fallback.
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int32, c_int64, c_size_t, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# TOIST_HIP_LIB: another build of the same library (kernel experiments: A/B two builds inside one GPU session)
LIB_PATH = os.environ.get("TOIST_HIP_LIB") or os.path.join(_HERE, "libtoist_hip.so")

TOIST_OK = 0
# operand kinds / activations (mirrors include/toist_hip.h)
A_ROWK, A_KROW, A_CONV, A_CONVT = 0, 1, 2, 3
B_ROWK, B_KROW, B_CONVX = 0, 1, 2
ACT_NONE, ACT_RELU, ACT_GELU, ACT_SIGMOID, ACT_MASK_POS, ACT_GELU_BWD, ACT_SIGMOID_BWD = range(7)


class Operand(Structure):
    _fields_ = [
        ("ptr", c_void_p), ("bs_outer", c_int64), ("bs_inner", c_int64), ("ld", c_int32), ("kin", c_int32),
        ("tap_stride", c_int64), ("SH", c_int32), ("SW", c_int32), ("SC", c_int32), ("PH", c_int32),
        ("PW", c_int32), ("R", c_int32), ("S", c_int32), ("stride", c_int32), ("pad", c_int32), ("dil", c_int32),
    ]


class Epilogue(Structure):
    _fields_ = [
        ("alpha", c_float), ("scale", c_void_p), ("shift", c_void_p), ("rscale", c_void_p), ("res", c_void_p), ("ldr", c_int32),
        ("aux", c_void_p), ("ldaux", c_int32), ("act", c_int32), ("pre_out", c_void_p), ("out_f32", c_int32),
        ("accumulate", c_int32), ("cmap", c_int32), ("cH", c_int32), ("cW", c_int32), ("cOH", c_int32),
        ("cOW", c_int32), ("cst", c_int32), ("res_div", c_int32), ("res_mod", c_int32), ("drop_where", c_int32), ("drop_p", c_float), ("drop_seed", c_uint64), ("drop_seed_dev", c_void_p),
    ]


class Gemm(Structure):
    _fields_ = [
        ("M", c_int32), ("N", c_int32), ("K", c_int32), ("a_kind", c_int32), ("b_kind", c_int32),
        ("a", Operand), ("b", Operand), ("c", c_void_p), ("ldc", c_int32), ("cs_outer", c_int64),
        ("cs_inner", c_int64), ("batch", c_int32), ("batch_inner", c_int32), ("split_k", c_int32),
        ("tile", c_int32), ("flags", c_int32), ("epi", Epilogue), ("workspace", c_void_p), ("a_colsum", c_void_p),
        ("group", c_void_p), ("a2", c_void_p), ("a2_from", c_int32),
    ]


ROW_PLAIN, ROW_LN_FWD, ROW_LN_BWD = 0, 1, 2


class RowGemm(Structure):
    """toist_rowgemm_desc (include/toist_hip.h): row-complete sub-layer launch of csrc/tlayer.hip"""
    _fields_ = [
        ("M", c_int32), ("K", c_int32), ("b_kind", c_int32), ("epi", c_int32), ("a", c_void_p), ("w", c_void_p), ("lda", c_int32), ("ldw", c_int32),
        ("fold", c_void_p), ("fold_stride", c_int64), ("fold_parts", c_int32), ("fold_cols", c_int32), ("bias", c_void_p), ("res", c_void_p),
        ("res2", c_void_p), ("ldr", c_int32), ("ldr2", c_int32), ("drop_p", c_float), ("eps", c_float), ("drop_seed", c_uint64),
        ("drop_seed_dev", c_void_p), ("gamma", c_void_p), ("beta", c_void_p), ("z", c_void_p), ("mean", c_void_p), ("rstd", c_void_p),
        ("out", c_void_p), ("ldo", c_int32), ("reserved", c_int32), ("add", c_void_p), ("out2", c_void_p), ("partials", c_void_p),
    ]


XDEC_MAX_LAYERS, XDEC_CTL_WORDS = 8, 1024


class XdecLayer(Structure):
    """toist_xdec_layer (include/toist_hip.h)"""
    _fields_ = [(n, c_void_p) for n in ("w_in", "b_in", "w_os", "b_os", "g1", "be1", "w_q", "b_q", "w_oc", "b_oc", "g3", "be3", "w1", "b1", "w2", "b2", "g4", "be4")] + \
               [("seed", c_uint64 * 6)]


class Xdec(Structure):
    """toist_xdec_desc (include/toist_hip.h): the XCD-resident decoder stack of csrc/xdec.hip"""
    _fields_ = [("B", c_int32), ("Q", c_int32), ("S", c_int32), ("L", c_int32), ("x0", c_void_p), ("qpos", c_void_p), ("kv", c_void_p), ("ldkv", c_int32),
                ("ff", c_int32), ("test_absent", c_int32), ("reserved", c_int32), ("key_pad", c_void_p), ("drop_p", c_float), ("eps", c_float), ("seed_dev", c_void_p)] + \
               [(n, c_void_p) for n in ("qkv", "ctx_s", "lse_s", "z1", "y1", "y1e", "mean1", "rstd1", "qc", "ctx_c", "lse_c", "z3", "y3", "mean3", "rstd3", "h", "z4", "y4",
                                        "y4e", "mean4", "rstd4", "part", "ctl", "prof")] + \
               [("layer", XdecLayer * XDEC_MAX_LAYERS)]


class XdecBwdLayer(Structure):
    """toist_xdec_bwd_layer (include/toist_hip.h)"""
    _fields_ = [(n, c_void_p) for n in ("w_in", "w_os", "w_q", "w_oc", "w1", "w2", "g1", "g3", "g4")] + [("seed", c_uint64 * 6)]


class XdecBwd(Structure):
    """toist_xdec_bwd_desc (include/toist_hip.h): backward of the XCD-resident decoder stack"""
    _fields_ = [("B", c_int32), ("Q", c_int32), ("S", c_int32), ("L", c_int32), ("kv", c_void_p), ("ldkv", c_int32), ("ldsink", c_int32), ("lddkv", c_int32),
                ("ff", c_int32), ("test_absent", c_int32), ("reserved", c_int32), ("key_pad", c_void_p), ("drop_p", c_float), ("reserved2", c_float), ("seed_dev", c_void_p)] + \
               [(n, c_void_p) for n in ("qkv", "ctx_s", "lse_s", "z1", "mean1", "rstd1", "qc", "ctx_c", "lse_c", "z3", "mean3", "rstd3", "h", "z4", "mean4", "rstd4", "g_out",
                                        "gb4", "dh", "go3", "go1", "sink", "dkv", "ln_part", "dctx", "part", "dq_part", "ctl", "prof")] + \
               [("layer", XdecBwdLayer * XDEC_MAX_LAYERS)]


class ReduceDesc(Structure):
    _fields_ = [("ws", c_void_p), ("out", c_void_p), ("rscale", c_void_p), ("splits", c_int32), ("M", c_int32), ("N", c_int32),
                ("ldc", c_int32), ("alpha", c_float), ("accumulate", c_int32)]


_SIGNATURES = {
    "toist_version": ([], ctypes.c_int),
    "toist_last_error": ([c_char_p, c_size_t], ctypes.c_int),
    "toist_matcher": ([c_void_p] * 6 + [c_int32] * 5 + [c_float] * 3 + [c_void_p] * 5, ctypes.c_int),
    "toist_gemm_bf16": ([POINTER(Gemm), c_void_p], ctypes.c_int),
    "toist_group_fill": ([c_void_p, c_int32, c_void_p, c_void_p], ctypes.c_int),
    "toist_layernorm_fwd": ([c_void_p, c_void_p, c_void_p, c_float, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p], ctypes.c_int),
    "toist_layernorm_bwd": ([c_void_p] * 5 + [c_int32, c_int32] + [c_void_p] * 4 + [c_float, c_uint64, c_void_p, c_void_p, c_int32, c_void_p], ctypes.c_int),
    "toist_layernorm_bwd_blocks": ([c_int32], ctypes.c_int),
    "toist_softmax_fwd": ([c_void_p, c_void_p] + [c_int32] * 5 + [c_void_p, c_void_p, c_float, c_uint64, c_void_p, c_void_p], ctypes.c_int),
    "toist_softmax_bwd": ([c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_float, c_uint64, c_void_p, c_void_p], ctypes.c_int),
    "toist_colsum": ([c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p], ctypes.c_int),
    "toist_add_bf16": ([c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p], ctypes.c_int),
    "toist_pack_image": ([c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p], ctypes.c_int),
    "toist_maxpool3x3s2": ([c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p], ctypes.c_int),
    "toist_stem_fwd": ([c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p], ctypes.c_int),
    "toist_unpack_nhwc": ([c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p], ctypes.c_int),
    "toist_stamp": ([c_void_p, c_int32, c_void_p], ctypes.c_int),
    "toist_text_prep": ([c_void_p, c_void_p, c_int32, c_int32, c_int64, c_void_p, c_void_p, c_void_p], ctypes.c_int),
    "toist_sine_position": ([c_void_p, c_int32, c_int32, c_int32, c_int32, c_float, c_void_p, c_void_p, c_void_p], ctypes.c_int),
    "toist_sine_position_seq": ([c_void_p, c_int32, c_int32, c_int32, c_int32, c_float, c_void_p, c_int32, c_void_p], ctypes.c_int),
    "toist_embed_fwd": ([c_void_p] * 5 + [c_int32, c_int32, c_void_p, c_void_p], ctypes.c_int),
    "toist_embed_bwd": ([c_void_p] * 3 + [c_int32, c_int32, c_int64] + [c_void_p] * 4, ctypes.c_int),
    "toist_criterion_fwd": ([c_void_p] * 9 + [c_int32] * 4 + [c_float, c_void_p, c_void_p, c_void_p], ctypes.c_int),
    "toist_criterion_bwd": ([c_void_p] * 9 + [c_int32] * 4 + [c_float, c_void_p, c_void_p, c_void_p, c_void_p], ctypes.c_int),
    "toist_contrastive_fwd": ([c_void_p] * 8 + [c_int32] * 5 + [c_float, c_void_p, c_void_p], ctypes.c_int),
    "toist_contrastive_bwd": ([c_void_p] * 8 + [c_int32] * 5 + [c_float, c_void_p, c_void_p, c_void_p, c_void_p], ctypes.c_int),
    "toist_l2norm_fwd": ([c_void_p, c_int32, c_int32, c_void_p, c_void_p], ctypes.c_int),
    "toist_l2norm_bwd": ([c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p], ctypes.c_int),
    "toist_attnmap_softmax_fwd": ([c_void_p, c_void_p] + [c_int32] * 5 + [c_void_p, c_void_p], ctypes.c_int),
    "toist_attnmap_softmax_bwd": ([c_void_p, c_void_p] + [c_int32] * 4 + [c_void_p, c_void_p], ctypes.c_int),
    "toist_groupnorm_apply": ([c_void_p] * 4 + [c_int32] * 4 + [c_float, c_int32, c_void_p, c_void_p], ctypes.c_int),
    "toist_groupnorm_fwd": ([c_void_p] * 3 + [c_int32] * 4 + [c_float, c_int32, c_void_p, c_void_p, c_void_p], ctypes.c_int),
    "toist_groupnorm_bwd": ([c_void_p] * 6 + [c_int32] * 4 + [c_float, c_int32] + [c_void_p] * 5, ctypes.c_int),
    "toist_upsample_add": ([c_void_p, c_void_p] + [c_int32] * 5 + [c_void_p, c_void_p], ctypes.c_int),
    "toist_upsample_add_bwd": ([c_void_p] + [c_int32] * 4 + [c_void_p, c_void_p], ctypes.c_int),
    "toist_sum_queries": ([c_void_p, c_int32, c_int32, c_int64, c_void_p, c_void_p], ctypes.c_int),
    "toist_upsample_add_rows": ([c_void_p, c_void_p, c_void_p] + [c_int32] * 5 + [c_void_p, c_void_p], ctypes.c_int),
    "toist_resize_add": ([c_void_p, c_void_p, c_void_p] + [c_int32] * 7 + [c_void_p, c_void_p], ctypes.c_int),
    "toist_resize_add_bwd": ([c_void_p] + [c_int32] * 6 + [c_void_p, c_void_p], ctypes.c_int),
    "toist_mask_stage_fwd": ([c_void_p] * 9 + [c_int32] * 9 + [c_float, c_void_p], ctypes.c_int),
    "toist_sum_segments": ([c_void_p, c_void_p, c_int32, c_int32, c_int64, c_void_p, c_void_p], ctypes.c_int),
    "toist_mask_loss_fwd": ([c_void_p] * 4 + [c_int32] * 5 + [c_float, c_void_p, c_void_p, c_void_p], ctypes.c_int),
    "toist_mask_loss_bwd": ([c_void_p] * 4 + [c_int32] * 5 + [c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p], ctypes.c_int),
    "toist_mask_loss_bwd_compact": ([c_void_p] * 4 + [c_int32] * 5 + [c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p], ctypes.c_int),
    "toist_dropout_bf16": ([c_void_p, c_int64, c_float, c_uint64, c_void_p, c_void_p, c_void_p], ctypes.c_int),
    "toist_attn2_splits": ([c_int32], ctypes.c_int),
    "toist_attn2_fwd": ([c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_int32, c_void_p] + [c_int32] * 5 + [c_float, c_float, c_uint64, c_void_p, c_void_p,
                        c_int32, c_void_p, c_void_p], ctypes.c_int),
    "toist_attn2_bwd": ([c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_void_p] + [c_int32] * 5 +
                        [c_float, c_float, c_uint64, c_void_p, c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_void_p], ctypes.c_int),
    "toist_rowgemm_blocks": ([c_int32], ctypes.c_int),
    "toist_rowgemm": ([POINTER(RowGemm), c_void_p], ctypes.c_int),
    "toist_xdec_supported": ([c_int32] * 4, ctypes.c_int),
    "toist_xdec_fwd": ([POINTER(Xdec), c_void_p], ctypes.c_int),
    "toist_xdec_bwd": ([POINTER(XdecBwd), c_void_p], ctypes.c_int),
    "toist_scatter_rows_f32": ([c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p], ctypes.c_int),
    "toist_kmeans": ([c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_int32, c_int32, c_int32, c_float, c_int32,
                     c_void_p, c_void_p, c_void_p, c_void_p], ctypes.c_int),
    "toist_attn_small_fwd": ([c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_int32, c_void_p] + [c_int32] * 4 + [c_float, c_float, c_uint64, c_void_p, c_void_p,
                             c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p], ctypes.c_int),
    "toist_attn_small_bwd": ([c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_int32, c_void_p] + [c_int32] * 4 + [c_float, c_float, c_uint64, c_void_p, c_void_p,
                             c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p], ctypes.c_int),
    "toist_conv3x3_small": ([c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p] + [c_int32] * 7 + [c_void_p], ctypes.c_int),
    "toist_wgrad3x3_small_blocks": ([], ctypes.c_int),
    "toist_wgrad3x3_small": ([c_void_p, c_void_p, c_void_p] + [c_int32] * 5 + [c_void_p], ctypes.c_int),
    "toist_lsap": ([c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p], ctypes.c_int),
    "toist_gemm_effective_split": ([POINTER(Gemm)], ctypes.c_int),
    "toist_gemm_pick_tile": ([POINTER(Gemm)], ctypes.c_int),
    "toist_splitk_reduce_batch": ([c_void_p, c_int32, c_void_p], ctypes.c_int),
    "toist_opt_chunk_elems": ([], ctypes.c_int),
    "toist_opt_sqnorm": ([c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p], ctypes.c_int),
    "toist_opt_finish_norm": ([c_void_p, c_int32, c_float, c_float, c_float, c_void_p, c_void_p], ctypes.c_int),
    "toist_opt_adamw_ema": ([c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_float, c_float, c_float, c_float, c_void_p], ctypes.c_int),
    "toist_opt_adamw_ema_blocks": ([c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_float, c_float, c_float, c_float, c_int32, c_void_p], ctypes.c_int),
    "toist_mask_resize_pack": ([c_void_p] + [c_int32] * 9 + [c_float, c_void_p, c_void_p], ctypes.c_int),
    "toist_mask_pack": ([c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p], ctypes.c_int),
    "toist_mask_unpack": ([c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p], ctypes.c_int),
    "toist_mask_area": ([c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p], ctypes.c_int),
    "toist_mask_iou": ([c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p], ctypes.c_int),
    "toist_mask_rle_count": ([c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p], ctypes.c_int),
    "toist_mask_rle_emit": ([c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p], ctypes.c_int),
    "toist_coco_match": ([c_void_p] * 8 + [c_int32, c_void_p, c_int32, c_void_p, c_int32] + [c_void_p] * 5, ctypes.c_int),
    "toist_mask_rle_counts": ([c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p], ctypes.c_int),
}

_lib = None


def exported_symbols():
    """Names every include/toist_hip.h entry point must be exported under."""
    return sorted(_SIGNATURES)


def lib():
    global _lib
    if _lib is None:
        # torch ships its own HIP runtime (libamdhip64); it must be the one already loaded when our
        # library binds, otherwise two runtimes coexist and launches fail with "no ROCm-capable device".
        import torch  # noqa: F401
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(toist_amd has no CPU / eager fallback)"
            )
        handle = ctypes.CDLL(LIB_PATH)
        for name, (argtypes, restype) in _SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the symbol is not exported
            fn.argtypes = argtypes
            fn.restype = restype
        _lib = handle
    return _lib


def last_error():
    buf = ctypes.create_string_buffer(512)
    lib().toist_last_error(buf, 512)
    return buf.value.decode("utf-8", "replace")


def check(rc, what):
    if rc != TOIST_OK:
        raise RuntimeError(f"{what} failed (code {rc}): {last_error()}")
