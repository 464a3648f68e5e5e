// Contrastive-alignment loss of TOIST/MDETR on the device, forward and backward, for every decoder layer in one launch.
//
// Replaces SetCriterion.loss_contrastive_align (/root/reference/models/mdetr.py:601-666): the reference builds a boolean
// [B, Q, L] query-to-token map on the HOST from the matcher's indices (a Python loop over images, matched pairs and
// character spans), copies it to the device and evaluates a two-sided InfoNCE with ~25 small tensor kernels per layer.
// Here the span of every TARGET is uploaded once per batch as a token bit mask (`tok_mask`, 2 x 64 bits per target row:
// the spans do not depend on the assignment), and one workgroup per (layer, image) turns the device-resident assignment
// of toist_matcher into the query -> token map, forms logits = pq . pt^T / temperature in fp32 and reduces
//   box -> token:  sum over queries with a positive of ( -sum_pos logit / (n_pos + 1e-6) + logsumexp_t logit )
//   token -> box:  the same over tokens (columns),
// exactly the expressions of mdetr.py:646-664.  MODE 1 re-forms the logits and emits d/d(proj_queries), d/d(proj_tokens)
// (the latter summed over the layers with fp32 atomics).  Tiny, latency-bound work (Q x L x 64 MACs per workgroup):
// neither roofline applies.  toist_l2norm_fwd/bwd are F.normalize(p=2, dim=-1) (mdetr.py:429-433) on fp32 rows.
#include <mutex>

#include "common.h"

namespace toist {

static constexpr int CA_THREADS = 1024;   // 16 waves: the loops below are chains of LDS reads per thread -- four waves per SIMD hide their latency (round 6: 28.9 + 44.6 -> 28.2 + 35.0 us)
static constexpr int CA_MW = 4;          // 64-bit mask words per target row
static constexpr int CA_MAX_TOK = 64 * CA_MW;   // 256 = the reference's max_text_len (models/mdetr.py:601-666 pads captions to it at most)
static constexpr int CA_LDS_MAX = 160 * 1024 - 256;   // dynamic LDS the kernels may ask for (they also hold a few static words)
static constexpr int CA_STAGE_TOK = 128; // token projections are staged in LDS up to this many tokens, read from L2 beyond (LDS: Q x T f32 logits)

__device__ __forceinline__ float ca_block_sum(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < CA_THREADS / 64; ++i) t += red[i];
    return t;
}

template <int MODE>
__global__ __launch_bounds__(CA_THREADS) void contrastive_kernel(
    const float* __restrict__ pq,            // [L,B,Q,D] normalised query projections
    const float* __restrict__ pt,            // [B,T,D]   normalised token projections
    const unsigned long long* __restrict__ tok_mask,   // [sum targets, CA_MW] token bits of every target row
    const int* __restrict__ tgt_off, const int* __restrict__ match_off, const long long* __restrict__ src_idx,
    const long long* __restrict__ tgt_idx, const float* __restrict__ num_boxes, int L, int B, int Q, int T, int D,
    float inv_temp,
    float* __restrict__ losses,              // MODE 0: [L] (+=)
    const float* __restrict__ upstream,      // MODE 1: [L]
    float* __restrict__ dpq,                 // MODE 1: [L,B,Q,D] fully written
    float* __restrict__ dpt) {               // MODE 1: [B,T,D] (+=, caller zeroes)
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int DP = D + 1;
    const bool staged = T <= CA_STAGE_TOK;
    const int TP = staged ? DP : D;          // row pitch of the token projections as read below
    unsigned long long* qm = reinterpret_cast<unsigned long long*>(sm);   // [Q][CA_MW] token bits of the target matched to query q (0 = unmatched)
    float* sq = sm + 2 * CA_MW * Q;          // [Q][DP]
    float* st_lds = sq + Q * DP;             // [T][DP] when staged
    float* lg = st_lds + (staged ? T * DP : 0);   // [Q][T]  logits, then d(logits)
    float* row_lse = lg + Q * T;             // [Q]
    float* col_lse = row_lse + Q;            // [T]
    float* col_np = col_lse + T;             // [T] positives per token
    __shared__ float red[CA_THREADS / 64];
    const int lb = blockIdx.x, l = lb / B, b = lb - l * B, tid = threadIdx.x;
    const int Mtot = match_off[B];

    for (int i = tid; i < CA_MW * Q; i += CA_THREADS) qm[i] = 0ull;
    const float* gq = pq + (size_t)(l * B + b) * Q * D;
    const float* gt = pt + (size_t)b * T * D;
    for (int i = tid; i < Q * D; i += CA_THREADS) sq[(i / D) * DP + (i % D)] = gq[i];
    if (staged)
        for (int i = tid; i < T * D; i += CA_THREADS) st_lds[(i / D) * DP + (i % D)] = gt[i];
    const float* const st = staged ? st_lds : gt;
    __syncthreads();
    for (int m = match_off[b] + tid; m < match_off[b + 1]; m += CA_THREADS) {
        const int q = (int)src_idx[(size_t)l * Mtot + m];
        const size_t t = (size_t)tgt_off[b] + (size_t)tgt_idx[(size_t)l * Mtot + m];
#pragma unroll
        for (int w = 0; w < CA_MW; ++w) qm[CA_MW * q + w] = tok_mask[CA_MW * t + w];          // a query is matched to at most one target
    }
    for (int i = tid; i < Q * T; i += CA_THREADS) {
        const int q = i / T, t = i - q * T;
        const float* a = sq + q * DP;
        const float* c = st + t * TP;
        float acc = 0.f;
        for (int d = 0; d < D; ++d) acc += a[d] * c[d];
        lg[i] = acc * inv_temp;
    }
    __syncthreads();

    auto bit = [&](int q, int t) -> bool { return (qm[CA_MW * q + (t >> 6)] >> (t & 63)) & 1ull; };
    float part = 0.f;
    // box -> token (rows)
    for (int q = tid; q < Q; q += CA_THREADS) {
        const float* r = lg + q * T;
        float mx = -INFINITY;
        for (int t = 0; t < T; ++t) mx = fmaxf(mx, r[t]);
        float se = 0.f, pos = 0.f;
        int np = 0;
        for (int t = 0; t < T; ++t) {
            se += __expf(r[t] - mx);
            if (bit(q, t)) { pos += r[t]; ++np; }
        }
        const float lse = mx + __logf(se);
        row_lse[q] = lse;
        if (np > 0) part += -pos / ((float)np + 1e-6f) + lse;
    }
    // token -> box (columns)
    for (int t = tid; t < T; t += CA_THREADS) {
        float mx = -INFINITY;
        for (int q = 0; q < Q; ++q) mx = fmaxf(mx, lg[q * T + t]);
        float se = 0.f, pos = 0.f;
        int np = 0;
        for (int q = 0; q < Q; ++q) {
            const float v = lg[q * T + t];
            se += __expf(v - mx);
            if (bit(q, t)) { pos += v; ++np; }
        }
        const float lse = mx + __logf(se);
        col_lse[t] = lse;
        col_np[t] = (float)np;
        if (np > 0) part += -pos / ((float)np + 1e-6f) + lse;
    }
    const float inv_nb = 1.f / num_boxes[0];
    if (MODE == 0) {
        const float tot = ca_block_sum(part, red);
        if (tid == 0) atomicAdd(losses + l, 0.5f * tot * inv_nb);
        return;
    }
    __syncthreads();
    // d(logits): rows with a positive contribute softmax_t - pm / n_row, columns with a positive softmax_q - pm / n_col
    const float coef = 0.5f * upstream[l] * inv_nb * inv_temp;
    for (int i = tid; i < Q * T; i += CA_THREADS) {
        const int q = i / T, t = i - q * T;
        const float v = lg[i];
        int npr = 0;
#pragma unroll
        for (int w = 0; w < CA_MW; ++w) npr += __popcll(qm[CA_MW * q + w]);
        const float pm = bit(q, t) ? 1.f : 0.f;
        float g = 0.f;
        if (npr > 0) g += __expf(v - row_lse[q]) - pm / ((float)npr + 1e-6f);
        if (col_np[t] > 0.f) g += __expf(v - col_lse[t]) - pm / (col_np[t] + 1e-6f);
        lg[i] = g * coef;
    }
    __syncthreads();
    float* oq = dpq + (size_t)(l * B + b) * Q * D;
    for (int i = tid; i < Q * D; i += CA_THREADS) {
        const int q = i / D, d = i - q * D;
        float acc = 0.f;
        for (int t = 0; t < T; ++t) acc += lg[q * T + t] * st[t * TP + d];
        oq[i] = acc;
    }
    float* ot = dpt + (size_t)b * T * D;
    for (int i = tid; i < T * D; i += CA_THREADS) {
        const int t = i / D, d = i - t * D;
        float acc = 0.f;
        for (int q = 0; q < Q; ++q) acc += lg[q * T + t] * sq[q * DP + d];
        atomicAdd(ot + i, acc);
    }
}

static size_t contrastive_lds(int Q, int T, int D) {
    const size_t f = (size_t)2 * CA_MW * Q + (size_t)Q * (D + 1) + (T <= CA_STAGE_TOK ? (size_t)T * (D + 1) : 0) + (size_t)Q * T + Q + 2 * (size_t)T;
    return f * sizeof(float);
}

// F.normalize(x, p=2, dim=-1, eps=1e-12) on fp32 rows: one wavefront per row.
// MODE 0: y = x / max(||x||, eps).  MODE 1: dx = (dy - y (y . dy)) / max(||x||, eps).
template <int MODE>
__global__ __launch_bounds__(256) void l2norm_kernel(const float* __restrict__ x, const float* __restrict__ dy, int rows, int D,
                                                     float* __restrict__ out) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* xr = x + (size_t)row * D;
    float ss = 0.f;
    for (int d = lane; d < D; d += 64) ss += xr[d] * xr[d];
    ss = wave_sum(ss);
    const float inv = 1.f / fmaxf(sqrtf(ss), 1e-12f);
    if (MODE == 0) {
        for (int d = lane; d < D; d += 64) out[(size_t)row * D + d] = xr[d] * inv;
    } else {
        const float* gr = dy + (size_t)row * D;
        float dot = 0.f;
        for (int d = lane; d < D; d += 64) dot += xr[d] * inv * gr[d];
        dot = wave_sum(dot);
        for (int d = lane; d < D; d += 64) out[(size_t)row * D + d] = (gr[d] - xr[d] * inv * dot) * inv;
    }
}

}  // namespace toist

using namespace toist;

static int contrastive_launch_ok(int L, int B, int Q, int T, int D, size_t* lds) {
    TOIST_REQUIRE(L > 0 && B > 0 && Q > 0 && T > 0 && D > 0, "toist_contrastive: bad shape L=%d B=%d Q=%d T=%d D=%d", L, B, Q, T, D);
    TOIST_REQUIRE(T <= CA_MAX_TOK, "toist_contrastive: %d tokens (the token masks hold %d)", T, CA_MAX_TOK);
    *lds = contrastive_lds(Q, T, D);
    TOIST_REQUIRE(*lds <= CA_LDS_MAX, "toist_contrastive: Q=%d T=%d D=%d needs %zu B of LDS (> %d)", Q, T, D, *lds, CA_LDS_MAX);
    if (*lds > 64 * 1024) {
        static std::atomic<unsigned long long> done{0};     // one bit per device
        if (!lds_attr_once_flag(done, [] {
                return hipFuncSetAttribute((const void*)contrastive_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, CA_LDS_MAX) == hipSuccess &&
                       hipFuncSetAttribute((const void*)contrastive_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, CA_LDS_MAX) == hipSuccess;
            })) {
            set_last_error("toist_contrastive: cannot enable %d bytes of LDS", CA_LDS_MAX);
            return TOIST_EHIP;
        }
    }
    return TOIST_OK;
}

extern "C" int toist_contrastive_fwd(const float* proj_queries, const float* proj_tokens, const uint64_t* tok_mask, const int32_t* tgt_off,
                                     const int32_t* match_off, const int64_t* src_idx, const int64_t* tgt_idx, const float* num_boxes, int L,
                                     int B, int Q, int T, int D, float temperature, float* losses, void* stream) {
    size_t lds;
    if (int rc = contrastive_launch_ok(L, B, Q, T, D, &lds)) return rc;
    TOIST_REQUIRE(temperature > 0.f, "toist_contrastive_fwd: temperature must be positive");
    hipLaunchKernelGGL(contrastive_kernel<0>, dim3(L * B), dim3(CA_THREADS), lds, (hipStream_t)stream, proj_queries, proj_tokens,
                       (const unsigned long long*)tok_mask, tgt_off, match_off, (const long long*)src_idx, (const long long*)tgt_idx, num_boxes, L, B,
                       Q, T, D, 1.f / temperature, losses, (const float*)nullptr, (float*)nullptr, (float*)nullptr);
    return check_launch("toist_contrastive_fwd");
}

extern "C" int toist_contrastive_bwd(const float* proj_queries, const float* proj_tokens, const uint64_t* tok_mask, const int32_t* tgt_off,
                                     const int32_t* match_off, const int64_t* src_idx, const int64_t* tgt_idx, const float* num_boxes, int L,
                                     int B, int Q, int T, int D, float temperature, const float* upstream, float* dproj_queries,
                                     float* dproj_tokens, void* stream) {
    size_t lds;
    if (int rc = contrastive_launch_ok(L, B, Q, T, D, &lds)) return rc;
    TOIST_REQUIRE(temperature > 0.f, "toist_contrastive_bwd: temperature must be positive");
    hipLaunchKernelGGL(contrastive_kernel<1>, dim3(L * B), dim3(CA_THREADS), lds, (hipStream_t)stream, proj_queries, proj_tokens,
                       (const unsigned long long*)tok_mask, tgt_off, match_off, (const long long*)src_idx, (const long long*)tgt_idx, num_boxes, L, B,
                       Q, T, D, 1.f / temperature, (float*)nullptr, upstream, dproj_queries, dproj_tokens);
    return check_launch("toist_contrastive_bwd");
}

extern "C" int toist_l2norm_fwd(const float* x, int rows, int D, float* y, void* stream) {
    TOIST_REQUIRE(rows >= 0 && D > 0, "toist_l2norm_fwd: bad shape");
    if (rows == 0) return TOIST_OK;
    hipLaunchKernelGGL(l2norm_kernel<0>, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, (const float*)nullptr, rows, D, y);
    return check_launch("toist_l2norm_fwd");
}

extern "C" int toist_l2norm_bwd(const float* x, const float* dy, int rows, int D, float* dx, void* stream) {
    TOIST_REQUIRE(rows >= 0 && D > 0, "toist_l2norm_bwd: bad shape");
    if (rows == 0) return TOIST_OK;
    hipLaunchKernelGGL(l2norm_kernel<1>, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, dy, rows, D, dx);
    return check_launch("toist_l2norm_bwd");
}
