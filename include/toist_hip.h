/*
 * toist_hip.h -- C ABI of the MI355X-native (gfx950) TOIST/MDETR hot path.
 *
 * The reference (AIR-DISCOVER/TOIST) has no FFI/plugin interface of its own: its hot path is stock
 * torch ops called from Python.  This header is therefore the boundary a
 * maintainer binds from Python with ctypes: plain pointers and sizes, no torch
 * types.  Each entry point names the reference call site it replaces.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer owned by the caller unless marked "host";
 *    the library never allocates or frees device memory and keeps no mutable global state;
 *  - every entry point only enqueues work on `stream` (a hipStream_t passed as void*) and never
 *    synchronises with the host;
 *  - return value: TOIST_OK (0) or a negative TOIST_E* code; the message for the last error of the
 *    calling thread is available from toist_last_error();
 *  - bf16 tensors are raw uint16 (upper half of an IEEE fp32), activations are row-major with the
 *    channel / feature axis innermost (NHWC for feature maps, [tokens, d] for sequences).
 */
#ifndef TOIST_HIP_H
#define TOIST_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TOIST_OK 0
#define TOIST_EINVAL (-1) /* bad shape / alignment / unsupported combination */
#define TOIST_EHIP (-2)   /* a HIP runtime call or kernel launch failed */

#define TOIST_ABI_VERSION 1

/* the library is built with -fvisibility=hidden: these entry points are its whole export table */
#if defined(__GNUC__)
#define TOIST_API __attribute__((visibility("default")))
#else
#define TOIST_API
#endif

TOIST_API int toist_version(void);
/* copies the calling thread's last error message (NUL terminated) into buf; returns its length */
TOIST_API int toist_last_error(char* buf, size_t cap);

/* ------------------------------------------------------------------------------------------------
 * Hungarian matcher.  Replaces HungarianMatcher.forward, /root/reference/models/matcher.py:39-87
 * (softmax + class/L1/GIoU cost, `.cpu()` sync, per-image scipy.optimize.linear_sum_assignment),
 * for L decoder layers x B images in one launch.
 *   logits   [L,B,Q,K] f32      boxes    [L,B,Q,4] f32 (cxcywh)
 *   tgt_boxes[Ttot,4]  f32      pos_map  [Ttot,K]  f32
 *   tgt_off  [B+1] i32 prefix sums of T_b          match_off [B+1] i32 prefix sums of min(Q,T_b)
 *   max_T    host value, max_b T_b (sizes the LDS)
 *   src_idx / tgt_idx [L, match_off[B]] i64: for image b the slice [match_off[b], match_off[b+1])
 *            holds the matched query indices (ascending) and their target indices (image-local),
 *            exactly the pair linear_sum_assignment returns for the [Q,T_b] block.
 *   status   [L*B] i32: 0 ok, 1 = cost block holds NaN/-inf (SciPy: "matrix contains invalid numeric
 *            entries"), 2 = infeasible.
 *   cost_out optional [L, B*Q, Ttot] f32 (only the per-image diagonal blocks are written).
 */
TOIST_API int toist_matcher(const float* logits, const float* boxes, const float* tgt_boxes, const float* pos_map,
                  const int32_t* tgt_off, const int32_t* match_off, int L, int B, int Q, int K, int max_T,
                  float w_class, float w_bbox, float w_giou, int64_t* src_idx, int64_t* tgt_idx,
                  int32_t* status, float* cost_out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * bf16 MFMA GEMM family:  C[z][m][n] = epilogue( sum_k A[z][m][k] * B[z][n][k] ).
 * One descriptor covers nn.Linear forward/dgrad/wgrad (transformer.py:273-304,362-408, HF RoBERTa),
 * the attention batched products, and ResNet convolutions as im2col-free implicit GEMM on NHWC
 * activations (backbone.py:75 -> torchvision ResNet body; FrozenBatchNorm2d, backbone.py:48-58, is
 * the per-channel scale/shift of the epilogue).
 */
enum {
    TOIST_A_ROWK = 0,  /* A[m][k] at a + m*lda + k                                  (k contiguous) */
    TOIST_A_KROW = 1,  /* A[m][k] at a + k*lda + m                                  (m contiguous) */
    TOIST_A_CONV = 2,  /* m = output pixel (n,oy,ox), k = (r,s,c): gather from NHWC source         */
    TOIST_A_CONVT = 3  /* m = input pixel (n,iy,ix),  k = (r,s,co): transposed-conv gather (dgrad) */
};
enum {
    TOIST_B_ROWK = 0, /* B[n][k] at b + n*ldb + k */
    TOIST_B_KROW = 1, /* B[n][k] at b + (k % kin)*ldb + (k / kin)*tap_stride + n */
    TOIST_B_CONVX = 2 /* n = (r,s,c), k = output pixel: gather from NHWC source (conv wgrad) */
};
enum {
    TOIST_ACT_NONE = 0,
    TOIST_ACT_RELU = 1,
    TOIST_ACT_GELU = 2,        /* exact erf GELU */
    TOIST_ACT_SIGMOID = 3,
    TOIST_ACT_MASK_POS = 4,    /* v = aux > 0 ? v : 0           (ReLU backward)       */
    TOIST_ACT_GELU_BWD = 5,    /* v *= gelu'(aux)               (aux = pre-activation) */
    TOIST_ACT_SIGMOID_BWD = 6  /* v *= aux * (1 - aux)          (aux = sigmoid output) */
};

typedef struct toist_operand {
    const void* ptr;     /* bf16 */
    int64_t bs_outer;    /* element stride of batch index z / batch_inner */
    int64_t bs_inner;    /* element stride of batch index z % batch_inner */
    int32_t ld;          /* leading dimension in elements */
    int32_t kin;         /* B_KROW two-level k (0 = plain) */
    int64_t tap_stride;  /* B_KROW two-level k */
    /* gather geometry (CONV / CONVT / CONVX) */
    int32_t SH, SW, SC;  /* source tensor [N,SH,SW,SC] */
    int32_t PH, PW;      /* pixel space enumerated by the row (A) or k (B) index: [N,PH,PW] */
    int32_t R, S, stride, pad, dil;
} toist_operand;

typedef struct toist_epilogue {
    float alpha;          /* v = acc * alpha */
    const float* scale;   /* [N] f32 or NULL: v = v * scale[n] */
    const float* shift;   /* [N] f32 or NULL: v = v + shift[n] */
    const float* rscale;  /* [M] f32 or NULL: v = v * rscale[m] (FrozenBN scale of a conv wgrad row) */
    const void* res;      /* bf16 residual, same row map as C, or NULL */
    int32_t ldr;
    const void* aux;      /* bf16 operand of the *_BWD / MASK_POS activations */
    int32_t ldaux;
    int32_t act;          /* TOIST_ACT_* */
    void* pre_out;        /* optional bf16 copy of v before the activation (ld = ldc) */
    int32_t out_f32;      /* C is f32 instead of bf16 */
    int32_t accumulate;   /* C += v (requires out_f32); read-modify-write by the owning thread, no atomics */
    /* optional row scatter of C/res/aux: m = (n,oy,ox) in [*,cOH,cOW] -> ((n*cH + oy*cst)*cW + ox*cst) */
    int32_t cmap, cH, cW, cOH, cOW, cst;
    /* optional residual broadcast: res row = (m / res_div) * res_mod + (m % res_mod) when res_div > 0
       (one residual map shared by the res_div / res_mod consecutive row blocks, e.g. all queries of an image) */
    int32_t res_div, res_mod;
    /* dropout: where = 0 none, 1 = before the residual add, 2 = after the activation */
    int32_t drop_where;
    float drop_p;
    uint64_t drop_seed;
    const uint64_t* drop_seed_dev; /* optional device word added to drop_seed at run time, so a captured
                                      hipGraph draws a fresh mask on every replay */
} toist_epilogue;

/* Grouped launch: `batch` problems of one shape whose operands are not uniformly strided (the weight gradients of the
 * identical residual blocks of a ResNet stage: 23 x three GEMMs in layer 3, each too small to fill the chip on its own and
 * therefore split along K with a fold pass -- grouped, they run as one unsplit launch).  Entry z replaces the strided
 * batch addressing: A and B start at a / b, C (and res / aux / pre_out) is offset by c_off elements from toist_gemm.c,
 * epi.rscale by rscale_off and a_colsum by colsum_off elements.  Device memory, `batch` entries; batch_inner must be 1.
 * With split_k > 1 the workspace is [problem][k-slice][M][N] and the caller folds every problem with
 * toist_splitk_reduce_batch (TOIST_GEMM_DEFER_REDUCE). */
typedef struct toist_group {
    const void* a;
    const void* b;
    int64_t c_off;
    int64_t rscale_off;
    int64_t colsum_off;
    int64_t shift_off;        /* offset (elements) added to epi.scale / epi.shift: per-problem bias of grouped nn.Linear forwards */
} toist_group;                /* 48 bytes */

typedef struct toist_gemm {
    int32_t M, N, K;
    int32_t a_kind, b_kind;
    toist_operand a, b;
    void* c;
    int32_t ldc;
    int64_t cs_outer, cs_inner; /* batch strides of C (and res/aux/pre_out) */
    int32_t batch, batch_inner; /* z in [0,batch): outer = z / batch_inner, inner = z % batch_inner */
    int32_t split_k;            /* >= 1; > 1 needs out_f32 and `workspace`: every k-slice stores its raw f32
                                   partial tile there and a second kernel reduces them into C */
    int32_t tile;               /* 0 = auto; 64 = 64x64x32, 65 = 64x64x64, 128 = 128x128x32, 129 = 128x128x64,
                                   130 = 128x64x64, 132 = 128x32x64, 133 = 32x128x64, 134 = 64x128x64 (BM x BN x BK);
                                   131 = shared-halo 3x3 kernel, 135 = short-K panel kernel (K <= 256, row-major A; error
                                   if the call does not qualify); + 256 * ring slots (optional) */
    int32_t flags;              /* bit0: build K-strided fragments with ds_write_b16 instead of
                                   ds_read_b64_tr_b16 (validation fallback) */
    toist_epilogue epi;
    float* workspace;           /* split_k > 1: f32 scratch of >= split_k * M * N elements (caller-owned) */
    float* a_colsum;            /* optional, A_KROW only: a_colsum[m] += sum_k A[m][k]  (f32 atomics) --
                                   the bias gradient of nn.Linear falls out of the wgrad GEMM's A tiles */
    const toist_group* group;   /* optional (device memory, `batch` entries): grouped launch, see toist_group */
    const void* a2;             /* optional, A_ROWK only: output columns n >= a2_from read their A rows from a2 (same ld / batch
                                   strides) -- nn.MultiheadAttention's packed in_proj applied to two inputs in one launch:
                                   q, k from x + pos, v from x (transformer.py:293-297, 370-400); a2_from % tile columns == 0 */
    int32_t a2_from;
} toist_gemm;

TOIST_API int toist_gemm_bf16(const toist_gemm* desc, void* stream);
/* writes n <= 64 host-side entries into a device table (they travel as kernel arguments: graph-capturable, no staging buffer) */
TOIST_API int toist_group_fill(const toist_group* rows, int n, toist_group* table, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Row kernels (bf16 data, f32 statistics).
 *  layernorm: nn.LayerNorm call sites transformer.py:279-280,341-345,481 and HF RoBERTa.
 *    bwd accumulates dgamma/dbeta with f32 atomics (caller zeroes them) and can also emit
 *    dx_drop = dropout-masked dx (mask regenerated from (seed, row*D+col), see toist_dropout_bf16).
 *  softmax:  the masked softmax inside nn.MultiheadAttention (transformer.py:273,337-338):
 *    scores/probs are [nbatch*H, Sq, ld] with ld >= roundup8(Sk); key_pad [nbatch,Sk] u8 (1 = pad).
 *    p_drop (optional) receives dropout(p) for the PV product.
 */
TOIST_API int toist_layernorm_fwd(const void* x, const float* gamma, const float* beta, float eps, int rows, int D,
                        void* y, float* mean, float* rstd, const void* add, void* y2, void* stream);
/*  optional second output y2 = y + add (both bf16 [rows, D]): the `src + pos` / `tgt + query_pos` that the next attention
 *  feeds to its q / k projections (transformer.py:293, 366, 386) without a separate elementwise launch */
TOIST_API int toist_layernorm_bwd(const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma,
                        int rows, int D, void* dx, float* dgamma, float* dbeta, void* dx_drop, float drop_p,
                        uint64_t seed, const uint64_t* seed_dev, float* partials, int partial_blocks, void* stream);
/* `partials` (optional, instead of dgamma / dbeta): f32 [2][partial_blocks][D] per-block sums of dy*xhat and dy, written with plain
 * stores (no contended atomics) for the caller to fold -- toist_splitk_reduce_batch with splits = partial_blocks, M = 1, N = D.
 * partial_blocks = toist_layernorm_bwd_blocks(rows). */
TOIST_API int toist_layernorm_bwd_blocks(int rows);
TOIST_API int toist_softmax_fwd(const void* scores, const uint8_t* key_pad, int nbatch, int H, int Sq, int Sk, int ld,
                      void* p, void* p_drop, float drop_p, uint64_t seed, const uint64_t* seed_dev, void* stream);
TOIST_API int toist_softmax_bwd(const void* p, const void* dp, int rows, int Sk, int ld, void* ds, float drop_p,
                      uint64_t seed, const uint64_t* seed_dev, void* stream);
/* every dropout seed is `seed + (seed_dev ? *seed_dev : 0)`: seed_dev is an optional device word */
/* out[n] += sum_m g[m][n]  (bias gradient; out is f32, caller zeroes it) */
TOIST_API int toist_colsum(const void* g, int M, int N, int ld, float* out, void* stream);
/* out = a + b, b broadcast with period b_period elements (with_pos_embed, transformer.py:287-288) */
TOIST_API int toist_add_bf16(const void* a, const void* b, int64_t n, int64_t b_period, void* out, void* stream);
/* out = dropout(x): keep iff hash(seed, flat index) >= p*2^32, kept values scaled by 1/(1-p) */
TOIST_API int toist_dropout_bf16(const void* x, int64_t n, float p, uint64_t seed, const uint64_t* seed_dev, void* out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Backbone-side layout / pooling kernels (NHWC bf16).
 *  pack_image : f32 NCHW [N,C<=8,H,W] -> bf16 NHWC [N,H,W,8] (zero padded channels): the stem operand
 *  maxpool    : 3x3 / stride 2 / pad 1 (ResNet stem, reached through backbone.py:87-89)
 *  unpack     : bf16 NHWC [N,HW,C] -> f32 NCHW [N,C,HW] for API-edge feature maps
 */
TOIST_API int toist_pack_image(const float* nchw, int N, int C, int H, int W, void* nhwc8, void* stream);
TOIST_API int toist_maxpool3x3s2(const void* in, int N, int H, int W, int C, void* out, void* stream);
TOIST_API int toist_unpack_nhwc(const void* nhwc, int N, int HW, int C, float* nchw, void* stream);
/* RoBERTa position ids (cumsum(id != pad) * (id != pad) + pad) and key-padding bytes (attention_mask != 1) of a tokenized batch [B, L] in one launch
 * (HF create_position_ids_from_input_ids / the text branch of transformer.py:129-138) */
TOIST_API int toist_text_prep(const int64_t* ids, const int64_t* attention_mask, int B, int L, int64_t pad_id, int64_t* pos_ids, uint8_t* key_pad, void* stream);
/* diagnostic: slots[idx] = the device's constant-rate clock (100 MHz ticks) when everything ordered before this launch on `stream` is done
 * (dates the branches of a replayed hipGraph without a profiler: bench.py --stamps; no reference counterpart) */
TOIST_API int toist_stamp(uint64_t* slots, int idx, void* stream);

/* ResNet stem in one launch (torchvision ResNet.conv1 / bn1 / relu / maxpool as used by /root/reference/models/backbone.py:64-91; frozen there):
 * image f32 NCHW [N, C <= 8, H, W] -> conv 7x7 / stride 2 / pad 3 with `weight` bf16 [64][7][7][8] (KRSC, channels padded to 8, FrozenBN scale
 * folded in) + shift[64] (f32, may be NULL) + ReLU -> max-pool 3x3 / stride 2 / pad 1 -> out bf16 NHWC [N, PH, PW, 64],
 * OH = (H - 1) / 2 + 1, PH = (OH - 1) / 2 + 1 (likewise for the width).  Replaces toist_pack_image + a toist_gemm_bf16 gather + toist_maxpool3x3s2. */
TOIST_API int toist_stem_fwd(const float* image, const void* weight, const float* shift, int N, int C, int H, int W, void* out, void* stream);

/* PositionEmbeddingSine.forward (position_encoding.py:30-49; normalize=True, scale=2*pi):
 * mask [B,H,W] u8 (1 = padded pixel) -> bf16 [B, H*W, 2*num_pos_feats] tokens and/or f32
 * [B, 2*num_pos_feats, H, W] (either output may be NULL). */
TOIST_API int toist_sine_position(const uint8_t* mask, int B, int H, int W, int num_pos_feats, float temperature, void* out_tok,
                        float* out_nchw, void* stream);
/* the same encoding as the token-major bf16 rows of a [B, rows_per_image, 2F] sequence buffer, rows_per_image >= H*W: the rows behind an image's
 * H*W tokens (the caption tokens of the cross-modal encoder input) are zero (transformer.py:139 torch.zeros_like(text_memory_resized)) */
TOIST_API int toist_sine_position_seq(const uint8_t* mask, int B, int H, int W, int num_pos_feats, float temperature, void* out_tok, int rows_per_image,
                                      void* stream);

/* RoBERTa input embeddings (HF RobertaEmbeddings, called at transformer.py:130):
 * out[t] = word[ids[t]] + type0 + pos[pos_ids[t]] (f32 tables -> bf16 [n,D]); bwd scatter-adds the
 * bf16 gradient rows into the dense f32 table gradients with atomics (any of them may be NULL). */
TOIST_API int toist_embed_fwd(const int64_t* ids, const int64_t* pos_ids, const float* word, const float* pos, const float* type0,
                    int n, int D, void* out, void* stream);
TOIST_API int toist_embed_bwd(const void* g, const int64_t* ids, const int64_t* pos_ids, int n, int D, int64_t pad_id, float* dword,
                    float* dpos, float* dtype0, void* stream); /* rows whose id == pad_id get no gradient (padding_idx) */

/* ------------------------------------------------------------------------------------------------
 * Set-criterion losses for all decoder layers at once.  Replaces SetCriterion.loss_labels / loss_boxes
 * / loss_cardinality (/root/reference/models/mdetr.py:488-518, 805-825, 783-803) given the device-
 * resident assignment of toist_matcher (same src_idx/tgt_idx/match_off/tgt_off).
 *   num_boxes  device f32[1] (already all-reduced / clamped, mdetr.py:997-1001)
 *   fwd: losses [L,4] f32 += (loss_ce, loss_bbox, loss_giou, cardinality_error) -- caller zeroes it
 *   match_status  optional int32[L*B] written by toist_matcher: a non-zero entry (NaN / -inf cost block, where SciPy raises
 *                 ValueError at /root/reference/models/matcher.py:85) poisons that layer's four losses with NaN, so the
 *                 caller's finite-loss guard (/root/reference/engine.py:82-85) trips without a host sync per call
 *   bwd: upstream [L,4] = d(total)/d(loss); dlogits [L,B,Q,K], dboxes [L,B,Q,4] are fully written
 */
TOIST_API int toist_criterion_fwd(const float* logits, const float* boxes, const float* tgt_boxes, const float* pos_map,
                        const int32_t* tgt_off, const int32_t* match_off, const int64_t* src_idx, const int64_t* tgt_idx,
                        const float* num_boxes, int L, int B, int Q, int K, float eos_coef, float* losses,
                        const int32_t* match_status, void* stream);
TOIST_API int toist_criterion_bwd(const float* logits, const float* boxes, const float* tgt_boxes, const float* pos_map,
                        const int32_t* tgt_off, const int32_t* match_off, const int64_t* src_idx, const int64_t* tgt_idx,
                        const float* num_boxes, int L, int B, int Q, int K, float eos_coef, const float* upstream,
                        float* dlogits, float* dboxes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Contrastive-alignment loss for all decoder layers at once.  Replaces SetCriterion.loss_contrastive_align
 * (/root/reference/models/mdetr.py:601-666) given the device-resident assignment of toist_matcher.
 *   proj_queries [L,B,Q,D] f32, proj_tokens [B,T,D] f32: the L2-normalised projections (mdetr.py:429-433)
 *   tok_mask     uint64 [sum_i T_i, 4]: bit t of row r = token t belongs to a positive span of target r
 *                (what the reference derives on the host from tokens_positive + char_to_token, :614-643; T <= 256 = max_text_len)
 *   fwd: losses [L] f32 += loss_contrastive_align of layer l (caller zeroes);  temperature = --temperature_NCE
 *   bwd: upstream [L]; dproj_queries [L,B,Q,D] fully written; dproj_tokens [B,T,D] += (caller zeroes)
 * toist_l2norm_fwd/bwd: F.normalize(x, p=2, dim=-1) on fp32 rows and its backward (dx from x, dy).
 */
TOIST_API int toist_contrastive_fwd(const float* proj_queries, const float* proj_tokens, const uint64_t* tok_mask, const int32_t* tgt_off,
                          const int32_t* match_off, const int64_t* src_idx, const int64_t* tgt_idx, const float* num_boxes, int L,
                          int B, int Q, int T, int D, float temperature, float* losses, void* stream);
TOIST_API int toist_contrastive_bwd(const float* proj_queries, const float* proj_tokens, const uint64_t* tok_mask, const int32_t* tgt_off,
                          const int32_t* match_off, const int64_t* src_idx, const int64_t* tgt_idx, const float* num_boxes, int L,
                          int B, int Q, int T, int D, float temperature, const float* upstream, float* dproj_queries,
                          float* dproj_tokens, void* stream);
TOIST_API int toist_l2norm_fwd(const float* x, int rows, int D, float* y, void* stream);
TOIST_API int toist_l2norm_bwd(const float* x, const float* dy, int rows, int D, float* dx, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Segmentation branch (config 3), /root/reference/models/segmentation.py.
 *  attnmap_softmax: per-head softmax over HW of MHAttentionMap.forward (:262-273, `flatten(3)`).  scores [B,Q,H,ld] bf16
 *      (+ key_pad [B,HW] u8) -> probabilities channels-last [B*Q, HW, H] bf16; bwd returns dscores [B*Q,H,ld].
 *  groupnorm: torch.nn.GroupNorm(G, C) (+ReLU) on NHWC bf16 [N,HW,C] (:203-241); stats / bstats are f32 [N,G,2]
 *      scratch owned by the caller; bwd accumulates dgamma/dbeta with atomics (caller zeroes them).
 *  upsample_add: out[bq,Y,X,:] = fpn[bq/Q,Y,X,:] + in[bq,Y/2,X/2,:] (nearest 2x + shared FPN term); bwd = 2x2 sums.
 *  sum_queries: out[b,i] = sum_q in[b,q,i] (backward of the per-query broadcasts).
 *  mask_loss: bilinear (align_corners=False) upsample of pred[pred_row[t]] [h,w] f32 to [TH,TW], sigmoid focal
 *      (alpha, gamma=2) + dice sums against gt[gt_row[t]] u8 [TH,TW] (mdetr.py:827-853, segmentation.py:276-319):
 *      sums[t] += {focal, p*t, p, t}; bwd scatters coef[0]*dfocal + coef[1]*ddice into dpred (f32 atomics).
 *      pred_row[t] < 0 = an unused slot of a fixed-capacity pair table: skipped by both kernels (sums stay 0).
 *      mask_loss_bwd_compact writes pair t's gradient to row t of dpred_rows [T,h,w] instead of row pred_row[t] of a [B*Q,h,w] tensor:
 *      the loss (mdetr.py:827-853 `src_masks = outputs["pred_masks"][src_idx]`) touches the matched maps only, every other map's
 *      gradient is exactly zero, and the mask head's backward (per-map convolutions / GroupNorm(8, C) statistics, segmentation.py:203-241)
 *      then runs on those T maps alone.
 *  upsample_add_rows: upsample_add on a gathered subset of the maps: map i is map rows[i] of the batch (its FPN term: image rows[i] / Q).
 *  groupnorm_fwd with y == NULL computes the statistics only; groupnorm_apply normalises with statistics computed earlier.
 *  mask_stage_fwd: one launch per stage of MaskHeadSmallConv's tail (segmentation.py:223-240): the 3x3 convolution `lay` / `out_lay` whose
 *      INPUT is built on the way in from the previous convolution's raw output src [N,SH,SW,c_in]: gn_in = GroupNorm(8, c_in) + ReLU from
 *      src_stats [N,8,2] ({sum, sum of squares}) / gamma / beta; up = nearest 2x (SH = H/2), with the FPN term of `adapter(fpn) + up2(x)`
 *      taken out of the convolution by linearity: fpn_conv [N/Q,H,W,c_out] bf16 = lay(adapter(fpn)) of the map's image (bias included, the
 *      caller's one small convolution per image) is added in the epilogue.  w [w_rows,3,3,c_in] bf16, bias f32 or NULL.  c_out > 1: out
 *      [N,H,W,c_out] bf16 = raw convolution output, out_stats [N,8,2] its GroupNorm(8, c_out) sums (zeroed here, f32 atomics); c_out == 1
 *      (out_lay): out is f32 [N,H,W].  Shapes: (32,16,gn_in,up), (64,32,up), (16,1,gn_in).
 *  sum_segments: out[b,i] = sum over maps s in [seg[b], seg[b+1]) of in[s,i] -- sum_queries for maps packed image by image (seg on the device).
 */
TOIST_API int toist_attnmap_softmax_fwd(const void* scores, const uint8_t* key_pad, int B, int Q, int H, int HW, int ld, void* out, void* stream);
TOIST_API int toist_attnmap_softmax_bwd(const void* prob, const void* dprob, int BQ, int H, int HW, int ld, void* dscores, void* stream);
TOIST_API int toist_groupnorm_fwd(const void* x, const float* gamma, const float* beta, int N, int HW, int C, int G, float eps, int relu,
                        void* y, float* stats, void* stream);
TOIST_API int toist_groupnorm_apply(const void* x, const float* stats, const float* gamma, const float* beta, int N, int HW, int C, int G, float eps, int relu,
                        void* y, void* stream);
/* y may be NULL when beta is given: the ReLU mask (y > 0) is then re-derived from x, stats, gamma and beta with the forward's arithmetic */
TOIST_API int toist_groupnorm_bwd(const void* dy, const void* y, const void* x, const float* stats, const float* gamma, const float* beta, int N, int HW,
                        int C, int G, float eps, int relu, void* dx, float* dgamma, float* dbeta, float* bstats, void* stream);
TOIST_API int toist_upsample_add(const void* in, const void* fpn, int BQ, int Q, int H, int W, int C, void* out, void* stream);
TOIST_API int toist_upsample_add_rows(const void* in, const void* fpn, const int64_t* rows, int n, int Q, int H, int W, int C, void* out, void* stream);
TOIST_API int toist_upsample_add_bwd(const void* dout, int BQ, int H, int W, int C, void* din, void* stream);
/* The same three for ANY target size (/root/reference/models/segmentation.py:218, 225, 232: `cur_fpn + F.interpolate(x, size=cur_fpn.shape[-2:],
 * mode="nearest")` -- the FPN level's size is 2H x 2W only for image sides that are multiples of 32): out [n, OH, OW, C] = fpn[image] + in[n, src(Y), src(X)]
 * with torch's source index min(floor(dst * (float)in / out), in - 1); rows = NULL (map i is map i, n % Q == 0) or the gathered subset as in
 * toist_upsample_add_rows.  toist_resize_add_bwd: din[n, y, x] = sum of the dout pixels that read it. */
TOIST_API int toist_resize_add(const void* in, const void* fpn, const int64_t* rows, int n, int Q, int H, int W, int OH, int OW, int C, void* out, void* stream);
TOIST_API int toist_resize_add_bwd(const void* dout, int BQ, int H, int W, int OH, int OW, int C, void* din, void* stream);
TOIST_API int toist_sum_queries(const void* in, int B, int Q, int64_t per, void* out, void* stream);
TOIST_API int toist_mask_stage_fwd(const void* src, const float* src_stats, const float* gamma, const float* beta, const void* fpn_conv, const void* w,
                        const float* bias, void* out, float* out_stats, int N, int Q, int H, int W, int c_in, int c_out, int w_rows, int gn_in, int up,
                        float eps, void* stream);
TOIST_API int toist_sum_segments(const void* in, const int32_t* seg, int B, int rows, int64_t per, void* out, void* stream);
TOIST_API int toist_mask_loss_fwd(const float* pred, const int32_t* pred_row, const uint8_t* gt, const int32_t* gt_row, int T, int h, int w,
                        int TH, int TW, float alpha, float* sums, const int32_t* valid_hw, void* stream);
TOIST_API int toist_mask_loss_bwd(const float* pred, const int32_t* pred_row, const uint8_t* gt, const int32_t* gt_row, int T, int h, int w,
                        int TH, int TW, float alpha, const float* sums, const float* coef, float* dpred, const int32_t* valid_hw, void* stream);
TOIST_API int toist_mask_loss_bwd_compact(const float* pred, const int32_t* pred_row, const uint8_t* gt, const int32_t* gt_row, int T, int h, int w,
                        int TH, int TW, float alpha, const float* sums, const float* coef, float* dpred_rows, const int32_t* valid_hw, void* stream);
/* valid_hw (all three; NULL = everything): device int32 [4] = {VH, VW, hs, ws} for a batch that sits in the corner of a larger bucket (a captured graph's
 * static shape).  The reference resizes its [hs, ws] prediction to the batch's largest image [VH, VW] (mdetr.py:843) and averages loss_mask over
 * those pixels: prediction rows / columns [0, hs) x [0, ws) are mapped onto target pixels [0, VH) x [0, VW) (align_corners = False grid of that
 * pair of sizes), nothing else takes part, and the caller divides loss_mask by VH * VW.  gt keeps its row stride TW. */

/* ---- optimizer tail: clip_grad_norm_ + AdamW + EMA + bf16 compute-copy refresh in one multi-tensor pass ------------
 * Replaces engine.py:87-101 of the reference (torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm);
 * optimizer.step() with torch.optim.AdamW over the 3 parameter groups of main.py:351-392; update_ema of
 * util/optim.py:9-26).  All tables live in device memory and are owned by the caller; `grads[i]` is the device
 * address of tensor i's fp32 gradient (0 = no gradient this step: the tensor is only EMA-averaged).
 * Work is cut into chunks of toist_opt_chunk_elems() elements: chunks[2*b] = tensor index, chunks[2*b+1] = chunk
 * index inside that tensor.  State (step count, clip coefficient, bias corrections) stays on the device so a
 * captured hipGraph replays the tail unchanged; learning rates are read from `groups` at run time. */
typedef struct toist_opt_tensor {
    float* p;                 /* fp32 master parameter (or the EMA source for buffers / frozen parameters)            */
    float* m;                 /* exp_avg     (NULL: never updated by the optimizer)                                   */
    float* v;                 /* exp_avg_sq                                                                           */
    float* ema;               /* EMA copy of p, or NULL                                                               */
    uint16_t* w;              /* bf16 compute copy refreshed from the new p, or NULL                                  */
    const float* row_scale;   /* optional: w = bf16(p * row_scale[element / row_len]) (folded FrozenBatchNorm scale)  */
    int64_t numel;
    int32_t row_len;
    int32_t group;            /* index into groups[]                                                                  */
} toist_opt_tensor;           /* 64 bytes */

typedef struct toist_opt_group {
    float lr;
    float weight_decay;
} toist_opt_group;

typedef struct toist_opt_state {
    float clip_coef;          /* min(1, max_norm / (grad_norm + 1e-6)); 1 when max_norm <= 0                          */
    float grad_norm;          /* total 2-norm of all gradients before clipping                                        */
    float bias1;              /* 1 - beta1^step                                                                       */
    float bias2_sqrt;         /* sqrt(1 - beta2^step)                                                                 */
    int32_t step;             /* optimizer steps taken (incremented by toist_opt_finish_norm)                         */
    int32_t reserved[3];
} toist_opt_state;            /* 32 bytes */

/* ---- attention cores (head dim 32; csrc/attn2.hip; the first-generation toist_attn_fwd / toist_attn_bwd of rounds 1-4 were removed in round 5).  Operands: per-head column
 * slices of [B*S, ld*] bf16 buffers), flash-style only: nothing score-shaped is stored, the key count is unbounded.
 *   toist_attn2_fwd   ctx = dropout(softmax(scale q k^T + key padding)) v; lse (f32 [B*H, Sq, 2]) receives (maximum of the RAW dot
 *                     products of the row, 1 / sum of exp(scale (s - max))).  The keep mask of element (row = bh * Sq + q, key) is the
 *                     16-bit field key & 1 of pair_hash((row * round8(Sk) + key) >> 1) compared with round(drop_p * 65536).
 *   toist_attn2_bwd   key-owning backward: dk / dv are written once; workgroup x of a head owns keys [128 x, 128 x + 128).  With
 *                     toist_attn2_splits(Sk) == 1 dq is written directly; otherwise every split leaves ITS share of dq in
 *                     dq_part (bf16 [splits][B*Sq][H*32], already multiplied by scale) and the consumer adds the shares
 *                     (toist_rowgemm's fold prologue); dq may then be NULL. */
TOIST_API int toist_attn2_splits(int Sk);
TOIST_API int toist_attn2_fwd(const void* q, int ldq, const void* kmat, int ldk, const void* v, int ldv, const uint8_t* key_pad, int B, int H, int Sq,
                    int Sk, int dh, float scale, float drop_p, uint64_t seed, const uint64_t* seed_dev, void* ctx, int ldo, float* lse,
                    void* stream);
TOIST_API int toist_attn2_bwd(const void* q, int ldq, const void* kmat, int ldk, const void* v, int ldv, const void* ctx, int ldo, const void* dctx, int lddo,
                    const float* lse, const uint8_t* key_pad, int B, int H, int Sq, int Sk, int dh, float scale, float drop_p, uint64_t seed,
                    const uint64_t* seed_dev, void* dq, int lddq, void* dk, int lddk, void* dv, int lddv, void* dq_part, void* stream);

/* ---- row-complete transformer sub-layers (d_model = 256; csrc/tlayer.hip).  One launch computes out[m][0..255] for blocks of 16
 * rows over the whole reduction length K and finishes the rows on chip:
 *   TOIST_ROW_LN_FWD  x + dropout(sublayer(x)) followed by LayerNorm -- `norm1(src + dropout1(src2))`, `norm2(src + dropout2(src2))` of
 *                     /root/reference/models/transformer.py:297-303 and norm1 / norm3 / norm4 of :376-407:
 *                       z = bf16(res + dropout(a w^T + bias));  out = LayerNorm(z) gamma + beta;  out2 = bf16(out + add) (optional);
 *                       z, mean, rstd are kept for the backward pass;
 *   TOIST_ROW_LN_BWD  the data gradient a w (+ res + res2: gradients arriving over residual connections) is the gradient w.r.t. a
 *                     LayerNorm OUTPUT; the launch applies the LayerNorm backward (saved z / mean / rstd, gamma) and writes
 *                       out = dz (gradient of the LayerNorm input = of the residual stream), out2 = dropout-masked dz (the gradient
 *                       of the dropout(sublayer) term, same (seed, m * 256 + n) hash as the forward pass; optional),
 *                       partials[2][blocks][256] f32 = per-block sums of g * xhat and g over the block's rows (d gamma / d beta, folded
 *                       by toist_splitk_reduce_batch; blocks = toist_rowgemm_blocks(M)); NULL when gamma is frozen;
 *   TOIST_ROW_PLAIN   out = bf16(a w (+ bias) + res + res2).
 * w is the nn.Linear parameter's bf16 copy read in place: TOIST_B_ROWK = [256][K] rows (forward: y = a w^T), TOIST_B_KROW = [K][256]
 * rows (data gradient: da = dy w).  Fold prologue (fold_parts > 1): columns [0, fold_cols) of every A row are the f32 sum of fold_parts
 * bf16 slabs (slab s, row m, column c at fold + s * fold_stride + m * fold_cols + c) -- the key-split partial dQ of toist_attn2_bwd --
 * rounded once and WRITTEN BACK to a (the weight-gradient GEMM reads the folded rows later).
 * All pointers 16-byte aligned, leading dimensions multiples of 8 elements, K a multiple of 128 (<= 4096).  The caller owns every
 * buffer; nothing is allocated, nothing synchronises. */
enum { TOIST_ROW_PLAIN = 0, TOIST_ROW_LN_FWD = 1, TOIST_ROW_LN_BWD = 2 };

typedef struct toist_rowgemm_desc {
    int32_t M, K;             /* rows, reduction length; the output is [M, 256] */
    int32_t b_kind;           /* TOIST_B_ROWK or TOIST_B_KROW */
    int32_t epi;              /* TOIST_ROW_* */
    const void* a;            /* bf16 [M, K], row stride lda (written when folding) */
    const void* w;            /* bf16 weights, row stride ldw */
    int32_t lda, ldw;
    const void* fold;         /* bf16 slabs of the fold prologue, or NULL */
    int64_t fold_stride;      /* elements between slabs */
    int32_t fold_parts, fold_cols;
    const float* bias;        /* f32 [256] or NULL */
    const void* res;          /* bf16 [M, 256], row stride ldr, or NULL */
    const void* res2;         /* bf16 [M, 256], row stride ldr2, or NULL */
    int32_t ldr, ldr2;
    float drop_p;             /* LN_FWD: dropout of (a w^T + bias);  LN_BWD: the mask of out2 */
    float eps;                /* LN_FWD */
    uint64_t drop_seed;
    const uint64_t* drop_seed_dev;   /* optional device word added to drop_seed (graph replay) */
    const float* gamma;       /* f32 [256] */
    const float* beta;        /* f32 [256] (LN_FWD) */
    void* z;                  /* bf16 [M, 256] contiguous: LN_FWD output (optional), LN_BWD input */
    float* mean;              /* f32 [M]: LN_FWD output (optional), LN_BWD input */
    float* rstd;
    void* out;                /* bf16 [M, 256], row stride ldo */
    int32_t ldo, reserved;
    const void* add;          /* LN_FWD: bf16 [M, 256] contiguous constant added into out2 */
    void* out2;               /* bf16 [M, 256] contiguous (see above), or NULL */
    float* partials;          /* LN_BWD: f32 [2][blocks][256], or NULL */
} toist_rowgemm_desc;

TOIST_API int toist_rowgemm_blocks(int M);
TOIST_API int toist_rowgemm(const toist_rowgemm_desc* d, void* stream);

/* ---- XCD-resident decoder stack, forward (/root/reference/models/transformer.py:225-267 TransformerDecoder.forward, :362-408
 * TransformerDecoderLayer.forward_post): ALL L post-norm decoder layers of d_model 256 / 8 heads / dim_feedforward 2048 in ONE launch.
 * Images are independent from the first decoder layer to the last, and an MI355X has 8 XCDs of 32 CUs with one L2 each: the launch is
 * 256 workgroups of 512 threads (one per CU); every workgroup reads its XCC id from the hardware register, takes a ticket on that XCD's
 * counter, and the 32 workgroups that read the same id form the group of image b = xcc (+ 8, + 16 ... for B > 8).  A group exchanges
 * activations only through its own L2 (plain stores, `s_waitcnt vmcnt(0)`, an L2-scope atomic arrival counter polled by one lane, sc1
 * loads that bypass the reader's L1): 0.84 us per synchronisation point (profiles/r05_xcd_barrier.txt), four per layer:
 *   P1  q | k | v = [x + query_pos | x] W_in^T + b_in             column tiles over the 32 CUs                      -> barrier
 *   A   a CU owns 4 query rows: self-attention of its rows (wave = head), out_proj + dropout + residual + norm1, the cross-attention
 *       query projection, cross-attention into the pre-projected memory K / V (wave = head, flash over 128-key blocks), out_proj +
 *       dropout + residual + norm3 -- all inside the workgroup                                                     -> barrier
 *   P6  a CU owns 64 of the 2048 hidden units: ReLU(y3 W1^T + b1) with dropout, and ITS partial sum of linear2    -> barrier
 *   P7  the row owners add the 32 partials, bias, dropout, residual, norm4 (and y4 + query_pos for the next layer) -> barrier
 * Every tensor the existing backward launches read (csrc/attn2.hip toist_attn2_bwd, csrc/tlayer.hip toist_rowgemm LN_BWD, the grouped
 * weight gradients) is written with the layouts, (maximum, 1 / sum) statistics and dropout hashes of the per-op forward kernels, so
 * the backward pass is unchanged.  Limits: Q <= 128, S <= 512, L <= 8, the device must expose 8 XCDs x 32 CUs.
 * ctl: 1024 uint32 of caller scratch, zeroed once by the caller; the launch function re-zeroes words 0 .. 1022 (a memset node under
 * capture).  ctl[1023] is sticky: != 0 = a bounded spin expired in some launch (a group was not co-resident) and that launch's
 * results are invalid.  Such a launch also overwrites every layer output (y4) of the affected images with NaN (the backward launch: their
 * `sink` rows), so the failure reaches the losses / the gradient norm even if nobody reads the status word. */
enum { TOIST_XDEC_MAX_LAYERS = 8, TOIST_XDEC_CTL_WORDS = 1024 };

typedef struct toist_xdec_layer {
    const void* w_in;  const float* b_in;      /* self_attn.in_proj_weight bf16 [768, 256] / bias f32 [768] */
    const void* w_os;  const float* b_os;      /* self_attn.out_proj */
    const float* g1;   const float* be1;       /* norm1 */
    const void* w_q;   const float* b_q;       /* cross_attn_image.in_proj rows 0 .. 255 (the query projection) */
    const void* w_oc;  const float* b_oc;      /* cross_attn_image.out_proj */
    const float* g3;   const float* be3;       /* norm3 */
    const void* w1;    const float* b1;        /* linear1 bf16 [2048, 256] */
    const void* w2;    const float* b2;        /* linear2 bf16 [256, 2048] */
    const float* g4;   const float* be4;       /* norm4 */
    uint64_t seed[6];                          /* dropout seeds: self-attention, norm1 branch, cross-attention, norm3 branch, hidden, norm4 branch */
} toist_xdec_layer;

typedef struct toist_xdec_desc {
    int32_t B, Q, S, L;
    const void* x0;            /* bf16 [B*Q, 256]: tgt entering layer 0 (zeros in the reference) */
    const void* qpos;          /* bf16 [B*Q, 256]: query_pos broadcast over the batch */
    const void* kv;            /* bf16 [B*S, ldkv]: memory K (with pos) / V projections of layer l at columns l*512 / l*512 + 256 */
    int32_t ldkv;
    int32_t ff;                /* dim_feedforward of linear1 / linear2: the launch is compiled for 2048, anything else is refused (TOIST_EINVAL) */
    int32_t test_absent;       /* 0.  Failure-path tests only: that many workgroups per XCD leave at once, as if they were not co-resident */
    int32_t reserved;
    const uint8_t* key_pad;    /* [B, S] 1 = padding, or NULL */
    float drop_p, eps;
    const uint64_t* seed_dev;  /* optional device word added to every seed (graph replay) */
    /* stacked per-layer outputs, layer l at + l * B*Q * width elements */
    void* qkv;                 /* bf16 [L][B*Q][768] */
    void* ctx_s;               /* bf16 [L][B*Q][256] self-attention context */
    float* lse_s;              /* f32  [L][B*8][Q][2] */
    void* z1; void* y1; void* y1e; float* mean1; float* rstd1;          /* norm1: pre-norm sum, output, output + query_pos */
    void* qc;                  /* bf16 [L][B*Q][256] cross-attention queries */
    void* ctx_c;               /* bf16 [L][B*Q][256] */
    float* lse_c;              /* f32  [L][B*8][Q][2] */
    void* z3; void* y3; float* mean3; float* rstd3;                      /* norm3 */
    void* h;                   /* bf16 [L][B*Q][2048] dropout(relu(linear1)) */
    void* z4; void* y4; void* y4e; float* mean4; float* rstd4;          /* norm4: y4 = the layer output (tgt_stack), y4e = y4 + query_pos */
    void* part;                /* bf16 [B][32][128][256] scratch: linear2 partial sums */
    uint32_t* ctl;             /* TOIST_XDEC_CTL_WORDS words of scratch */
    uint64_t* prof;            /* diagnostics, normally NULL: [256 workgroups][L][16] device-clock stamps (100 MHz) at the phase boundaries of image 0 .. 7 */
    toist_xdec_layer layer[TOIST_XDEC_MAX_LAYERS];
} toist_xdec_desc;

TOIST_API int toist_xdec_supported(int B, int Q, int S, int L);      /* 1 when toist_xdec_fwd takes the shape on the current device */
TOIST_API int toist_xdec_fwd(const toist_xdec_desc* d, void* stream);

/* ---- XCD-resident decoder stack, backward: the data-gradient chain of ALL L decoder layers in one launch, consuming what toist_xdec_fwd saved
 * (same grouping: one image per XCD, 32 workgroups of 512 threads, XCD-local barriers; six per layer):
 *   R0  row owners: norm4 backward of their 4 rows (gradient of the layer output = next layer's input gradient + the shared final norm's share)
 *   H   CU = 64 hidden units: dh = (dzd4 W2) masked by h > 0, and ITS partial sum of dh W1
 *   C   row owners: fold the 32 partials + residual gradient, norm3 backward, gradient of the cross-attention context (x W_oc)
 *   D   CU = (head, key split): cross-attention backward (the body of toist_attn2_bwd: dk / dv of the memory written once, dq shares)
 *   E   row owners: fold dq, x W_q + residual gradient, norm1 backward, gradient of the self-attention context (x W_os)
 *   F   CU = head: self-attention backward (dq | dk | dv of the layer's q | k | v)
 *   G   row owners: [dq | dk | dv] W_in + residual gradient + the final norm's share of the layer below = the next R0's input
 * Weight gradients are NOT formed here: every operand they need (dzd4, dh, dzd3, dq, dzd1, dq | dk | dv) is left in the output tensors for the
 * grouped weight-gradient launches, and the LayerNorm parameter gradients leave as per-row-block partial sums (ln_part) for
 * toist_splitk_reduce_batch.  Limits as toist_xdec_fwd. */
typedef struct toist_xdec_bwd_layer {
    const void* w_in;          /* bf16 weights, read in place as k-major operands */
    const void* w_os;
    const void* w_q;
    const void* w_oc;
    const void* w1;
    const void* w2;
    const float* g1;           /* LayerNorm gammas */
    const float* g3;
    const float* g4;
    uint64_t seed[6];          /* the forward's seeds */
} toist_xdec_bwd_layer;

typedef struct toist_xdec_bwd_desc {
    int32_t B, Q, S, L;
    const void* kv;            /* as toist_xdec_desc */
    int32_t ldkv, ldsink, lddkv;
    int32_t ff;                /* as toist_xdec_desc: must be 2048 */
    int32_t test_absent;       /* as toist_xdec_desc: 0 outside failure-path tests */
    int32_t reserved;
    const uint8_t* key_pad;
    float drop_p, reserved2;
    const uint64_t* seed_dev;
    /* saved by the forward launch (stacked per layer) */
    const void* qkv; const void* ctx_s; const float* lse_s; const void* z1; const float* mean1; const float* rstd1;
    const void* qc; const void* ctx_c; const float* lse_c; const void* z3; const float* mean3; const float* rstd3;
    const void* h; const void* z4; const float* mean4; const float* rstd4;
    const void* g_out;         /* bf16 [L][B*Q][256]: gradient of the shared final norm w.r.t. every layer's output */
    /* outputs */
    void* gb4;                 /* bf16 [L][B*Q][256]: (dropout-masked) gradient of norm4's input = dy of linear2 */
    void* dh;                  /* bf16 [L][B*Q][2048]: gradient of linear1's output */
    void* go3;                 /* bf16 [L][B*Q][256]: dy of cross_attn out_proj */
    void* go1;                 /* bf16 [L][B*Q][256]: dy of self_attn out_proj */
    void* sink;                /* bf16 [B*Q][ldsink]: per layer l at columns l*1024: [dq_s | dk_s | dv_s | dq_c] */
    void* dkv;                 /* bf16 [B*S][lddkv]: dk / dv of the memory projections, layer l at columns l*512 / l*512 + 256 */
    float* ln_part;            /* f32 [L][3][2][B*ceil(Q/4)][256]: (sum v*xhat, sum v) per row block for norm1 / norm3 / norm4 */
    /* scratch */
    void* dctx;                /* bf16 [2][B*Q][256] */
    void* part;                /* bf16 [B][32][128][256] */
    void* dq_part;             /* bf16 [4][B*Q][256] */
    uint32_t* ctl;
    uint64_t* prof;
    toist_xdec_bwd_layer layer[TOIST_XDEC_MAX_LAYERS];
} toist_xdec_bwd_desc;

TOIST_API int toist_xdec_bwd(const toist_xdec_bwd_desc* d, void* stream);

/* ---- masked row scatter (the nearest-replacement memory-bank update, /root/reference/models/mdetr.py:98-103, when the pair table has a fixed
 * capacity and the live slots are only known on the device: hipGraph replay of the distillation step).  dst[dst_row[i]] = src[src_row[i]] (rows of d
 * floats) for every i < m with src_row[i] >= 0 and dst_row[i] >= 0; live destinations must be distinct. */
TOIST_API int toist_scatter_rows_f32(const float* src, const int64_t* src_row, float* dst, const int64_t* dst_row, int m, int d, void* stream);

/* ---- k-means of the distillation step on the device (models/kmeans.py:21-96 as mdetr.py:213-234 calls it).  One workgroup per
 * distinct task of the batch: group g covers samples members[group_off[g] .. group_off[g+1]) (batch order), all of task
 * group_task[g]; for each sample: Lloyd iterations over banks[task] ([N, D] f32, stride bank_stride elements) from
 * centers[task] ([K, D] f32, stride centers_stride, updated IN PLACE) until (sum_k |shift_k|)^2 < tol (or max_iter), then
 * pick[sample] = index of the centre nearest to features[sample] ([*, D] f32) and chosen_center[sample] = that centre.
 * No host synchronisation; iters (optional) receives the iteration count per sample. */
TOIST_API int toist_kmeans(const float* banks, int64_t bank_stride, float* centers, int64_t centers_stride, const int32_t* group_task,
                 const int32_t* group_off, const int32_t* members, int n_groups, const float* features, int N, int D, int K, float tol,
                 int max_iter, int32_t* pick, float* chosen_center, int32_t* iters, void* stream);

/* ---- whole-head self-attention for short sequences (S <= 64 keys, head dim <= 64, % 8 == 0): the text encoder's attention
 * (RoBERTa over a 16-token caption: transformer.py:129-130 via transformers.RobertaSelfAttention) as ONE launch each way instead of
 * batched 16 x 16 GEMMs + softmax kernels.  q / k / v / ctx / gradients are per-head column slices of [B*S, ld*] bf16 buffers
 * (row b*S + s, feature h*dh + e); bq / bk / bv (optional, f32 [H*dh]) are the projection biases, added on load so the packed
 * q | k | v projection needs no epilogue vector; key_pad [B, S] u8 (1 = padding) or NULL; stats f32 [B*H, S, 2] = (row maximum,
 * 1 / row sum) written by fwd, read by bwd, which re-forms the probabilities and the dropout mask ((seed + *seed_dev, element)). */
TOIST_API int toist_attn_small_fwd(const void* q, int ldq, const void* kmat, int ldk, const void* v, int ldv, const uint8_t* key_pad, int B, int H, int S,
                         int dh, float scale, float drop_p, uint64_t seed, const uint64_t* seed_dev, void* ctx, int ldo, float* stats,
                         const float* bq, const float* bk, const float* bv, void* stream);
TOIST_API int toist_attn_small_bwd(const void* q, int ldq, const void* kmat, int ldk, const void* v, int ldv, const uint8_t* key_pad, int B, int H, int S,
                         int dh, float scale, float drop_p, uint64_t seed, const uint64_t* seed_dev, const float* stats, const void* dctx,
                         int lddo, void* dq, int lddq, void* dk, int lddk, void* dv, int lddv, const float* bq, const float* bk,
                         const float* bv, void* stream);

/* ---- 3x3 / stride 1 / pad 1 convolution with <= 32 channels on either side (mask-head stages at 160x160,
 * segmentation.py:176-241 lay5 / out_lay; HBM-bound): NHWC bf16 in / out, weights [w_co][3][3][w_ci] bf16.
 * dgrad = 0: out[p, co] = shift[co] + sum x[p + tap, ci] w[co, tap, ci] (+ res);  c_src = w_ci, c_out = w_co.
 * dgrad = 1: out[p, ci] = sum dy[p - tap, co] w[co, tap, ci] (+ res);             c_src = w_co, c_out = w_ci. */
TOIST_API int toist_conv3x3_small(int dgrad, const void* src, const void* w, const float* shift, const void* res, void* out, int n_img, int H, int W,
                        int c_src, int c_out, int w_co, int w_ci, void* stream);

/* weight gradient of the same convolutions: every one of toist_wgrad3x3_small_blocks() workgroups writes an fp32 partial
 * [c_out][9 * c_in] to ws; fold them with toist_splitk_reduce_batch (splits = blocks, M = c_out, N = 9 * c_in). */
TOIST_API int toist_wgrad3x3_small_blocks(void);
TOIST_API int toist_wgrad3x3_small(const void* dy, const void* x, float* ws, int n_img, int H, int W, int c_in, int c_out, void* stream);

/* ---- batched linear sum assignment on caller-supplied cost matrices ---------------------------------------------
 * scipy.optimize.linear_sum_assignment for the reference's other call sites (mdetr.py:100 memory-bank replacement on
 * an L1 cdist, mdetr.py:539 softkd matcher): problem p is the row-major fp32 matrix [rows[p], cols[p]] at
 * cost + offset[p]; min(rows, cols) pairs are written at row_idx / col_idx + out_off[p], rows ascending (SciPy's
 * order).  status[p]: 0 ok, 1 NaN or -inf in the matrix (SciPy: ValueError), 2 infeasible. */
TOIST_API int toist_lsap(const float* cost, const int64_t* offset, const int32_t* rows, const int32_t* cols, int ld /* row stride, 0 = cols[p] */,
               int n, int max_rows, int max_cols,
               int64_t max_cells /* max over problems of rows*cols: sizes the LDS */, const int64_t* out_off, int64_t* row_idx, int64_t* col_idx, int32_t* status, void* stream);

/* ---- deferred split-K reductions ------------------------------------------------------------------------
 * toist_gemm_bf16 with flags bit1 (TOIST_GEMM_DEFER_REDUCE) writes the k-slice partials to `workspace` and returns
 * without launching the reduction; the caller later folds up to many such GEMMs with ONE launch per 48 descriptors:
 *   out[m][n] (+)= alpha * rscale[m] * sum_s ws[s][m][n].
 * toist_gemm_effective_split tells how many slices the kernel really wrote (k-slices that would own no k-tile are
 * dropped).  `descs` is a HOST array (copied into the kernel arguments); nothing is kept after the call returns. */
#define TOIST_GEMM_DEFER_REDUCE 2
/* flags bit2: split_k > 1 with the COMPLETE epilogue.  The k-slices store raw f32 partial tiles to `workspace` ([k-slice][M][N]) as
 * for any split, and a second kernel adds them in slice order and applies scale / shift / dropout / residual / activation / the
 * output type exactly as the un-split GEMM would -- for nn.Linear GEMMs with few output tiles and a deep reduction (128 tokens,
 * K = 3072: 24 tiles on 256 CUs).  batch = 1, no group, no row map; N %% 4 == 0. */
#define TOIST_GEMM_SPLIT_EPILOGUE 4
/* tile code the dispatcher would pick for this descriptor (`tile` = 0) */
TOIST_API int toist_gemm_pick_tile(const toist_gemm* desc);
typedef struct toist_reduce_desc {
    const float* ws;          /* [splits][M][N] fp32 partials                */
    float* out;               /* [M][ldc] fp32                               */
    const float* rscale;      /* optional per-row scale [M]                  */
    int32_t splits, M, N, ldc;
    float alpha;
    int32_t accumulate;       /* 1: out += ..., 0: out = ...                 */
} toist_reduce_desc;          /* 48 bytes */
TOIST_API int toist_gemm_effective_split(const toist_gemm* desc);
TOIST_API int toist_splitk_reduce_batch(const toist_reduce_desc* descs, int n, void* stream);

TOIST_API int toist_opt_chunk_elems(void);
TOIST_API int toist_opt_sqnorm(const toist_opt_tensor* table, const int64_t* grads, const int32_t* chunks, int n_chunks, float* partial, void* stream);
TOIST_API int toist_opt_finish_norm(const float* partial, int n_chunks, float max_norm, float beta1, float beta2, toist_opt_state* state, void* stream);
TOIST_API int toist_opt_adamw_ema(const toist_opt_tensor* table, const int64_t* grads, const int32_t* chunks, int n_chunks,
                        const toist_opt_group* groups, const toist_opt_state* state, float beta1, float beta2, float eps,
                        float ema_decay, void* stream);
/* the same update issued by at most max_blocks workgroups (0 = one per chunk): a slim launch that shares the chip with other work -- the
 * late parameter groups of toist_amd.optim, updated beside the next forward pass */
TOIST_API int toist_opt_adamw_ema_blocks(const toist_opt_tensor* table, const int64_t* grads, const int32_t* chunks, int n_chunks,
                               const toist_opt_group* groups, const toist_opt_state* state, float beta1, float beta2, float eps,
                               float ema_decay, int max_blocks, void* stream);

/* ---- evaluation masks (replaces PostProcessSegm's dense resizes, models/postprocessors.py:73-109, and what
 * datasets/coco_eval.py:307-332 asks of pycocotools: mask_util.encode, and maskUtils.iou / area inside COCOeval).
 * A binary mask [h, w] is a column-major bit plane: uint64 bits[mask][x][yw], yw < ceil(h/64), bit b of word yw = pixel
 * (y = 64*yw + b, x), bits beyond h zero -- RLE's pixel order (index x*h + y).  n <= 65535 masks per call.
 *   resize_pack : src [n, h0, w0] fp32 mask logits -> bilinear (align_corners false) to [max_h, max_w], its
 *                 [crop_h, crop_w] corner -> bilinear to [h, w], sigmoid(v) > threshold, packed
 *   pack/unpack : dense [n, h, w] bytes (non-zero = foreground; unpack writes 0/1)
 *   area        : foreground pixels per mask
 *   iou         : iou[d, g] (row-major double) = i / (iscrowd[g] ? area_dt[d] : area_dt[d] + area_gt[g] - i), 0 when i = 0
 *   rle_count   : column_transitions[m, x] = value changes inside column x, counted against the preceding pixel in RLE
 *                 order (the last pixel of column x-1; 0 before the first pixel)
 *   rle_emit    : positions[column_offset[m, x] ...] = pixel indices of those changes, ascending (column_offset = an
 *                 exclusive prefix sum of column_transitions over the flattened [n, w] array)
 *   rle_counts  : mask m owns positions[first_position[m] .. first_position[m+1]) and writes its
 *                 (transitions + 1) run lengths, zeros first, to counts[first_run[m] ...]  (= rleEncode's cnts) */
TOIST_API int toist_mask_resize_pack(const float* src, int n, int h0, int w0, int max_h, int max_w, int crop_h, int crop_w, int h, int w,
                           float threshold, uint64_t* bits, void* stream);
TOIST_API int toist_mask_pack(const uint8_t* dense, int n, int h, int w, uint64_t* bits, void* stream);
TOIST_API int toist_mask_unpack(const uint64_t* bits, int n, int h, int w, uint8_t* dense, void* stream);
TOIST_API int toist_mask_area(const uint64_t* bits, int n, int h, int w, uint32_t* area, void* stream);
TOIST_API int toist_mask_iou(const uint64_t* dt, int n_dt, const uint64_t* gt, int n_gt, const uint8_t* iscrowd, const uint32_t* area_dt,
                   const uint32_t* area_gt, int h, int w, double* iou, void* stream);
TOIST_API int toist_mask_rle_count(const uint64_t* bits, int n, int h, int w, int32_t* column_transitions, void* stream);
TOIST_API int toist_mask_rle_emit(const uint64_t* bits, int n, int h, int w, const int64_t* column_offset, uint32_t* positions, void* stream);
TOIST_API int toist_mask_rle_counts(const uint32_t* positions, const int64_t* first_position, const int64_t* first_run, int n, int h, int w,
                          uint32_t* counts, void* stream);

/* COCOeval.evaluateImg over a batch of images (pycocotools cocoeval.py; reached from datasets/coco_eval.py:368-399).  Image i owns
 * detections [dt_offset[i], dt_offset[i+1]) (score-descending, already cut to maxDets[-1]), ground truth [gt_offset[i], gt_offset[i+1])
 * and the row-major [D_i, G_i] block of `iou` at iou_offset[i].  For every area range a = [lo, hi] and IoU threshold t:
 *   dt_match [A*T*dt_offset[i] + (a*T + t)*D_i + d] = ground-truth index matched by detection d (within the image), or -1
 *   dt_ignore[same index]                          = matched an ignored ground truth, or unmatched with its area outside the range
 *   gt_range_ignore[A*gt_offset[i] + a*G_i + g]    = gt_ignore[g] or its area outside the range
 * gt_taken (A*T bytes per ground truth) is scratch.  Offsets tables have n_images + 1 entries. */
TOIST_API int toist_coco_match(const double* iou, const int64_t* iou_offset, const double* dt_area, const int64_t* dt_offset, const double* gt_area,
                     const uint8_t* gt_ignore, const uint8_t* gt_crowd, const int64_t* gt_offset, int n_images, const double* area_ranges,
                     int n_ranges, const double* iou_thresholds, int n_thresholds, int32_t* dt_match, uint8_t* dt_ignore,
                     uint8_t* gt_range_ignore, uint8_t* gt_taken, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TOIST_HIP_H */
