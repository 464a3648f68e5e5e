// Set-criterion losses of TOIST/MDETR, forward and backward, for every decoder layer in one launch.
//
// Replaces SetCriterion.loss_labels / loss_boxes / loss_cardinality
// (/root/reference/models/mdetr.py:488-518, 805-825, 783-803; GIoU from util/box_ops.py:40-61) -- about a
// hundred tiny tensor kernels per step in the reference -- with one workgroup per (layer, image):
// the assignment produced by matcher.hip is turned into a query -> target map in LDS, each wave then
// handles query rows (log-softmax over K, soft-target cross entropy, eos weighting, L1 + GIoU of matched
// pairs) and a block reduction adds the image's contribution to losses[layer][0..3].
// Tiny, latency-bound work (0.8 MB of logits per layer): neither roofline applies.
#include "common.h"

namespace toist {

static constexpr int CRIT_THREADS = 1024;   // 16 waves, one query per wave at a time (the kernel is latency-bound: 48 blocks)

__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < CRIT_THREADS / 64; ++i) t += red[i];
    return t;
}

struct GiouPair {
    float giou;
    float g[4];  // d(giou)/d(cx, cy, w, h) of the prediction
};

// box_ops.generalized_box_iou restricted to one (prediction, target) pair, with its gradient
__device__ __forceinline__ GiouPair giou_pair(const float* a, const float* b) {
    const float ax0 = a[0] - 0.5f * a[2], ay0 = a[1] - 0.5f * a[3], ax1 = a[0] + 0.5f * a[2], ay1 = a[1] + 0.5f * a[3];
    const float bx0 = b[0] - 0.5f * b[2], by0 = b[1] - 0.5f * b[3], bx1 = b[0] + 0.5f * b[2], by1 = b[1] + 0.5f * b[3];
    const float aw = ax1 - ax0, ah = ay1 - ay0;
    const float area_a = aw * ah, area_b = (bx1 - bx0) * (by1 - by0);
    const float iw_raw = fminf(ax1, bx1) - fmaxf(ax0, bx0), ih_raw = fminf(ay1, by1) - fmaxf(ay0, by0);
    const float iw = fmaxf(iw_raw, 0.f), ih = fmaxf(ih_raw, 0.f);
    const float inter = iw * ih;
    const float uni = (area_a + area_b) - inter;
    const float iou = inter / uni;
    const float ew_raw = fmaxf(ax1, bx1) - fminf(ax0, bx0), eh_raw = fmaxf(ay1, by1) - fminf(ay0, by0);
    const float ew = fmaxf(ew_raw, 0.f), eh = fmaxf(eh_raw, 0.f);
    const float hull = ew * eh;
    GiouPair r;
    r.giou = iou - (hull - uni) / hull;
    // reverse mode through the expressions above
    const float g_union = -inter / (uni * uni) + 1.f / hull;
    const float g_hull = -uni / (hull * hull);
    const float g_inter = 1.f / uni - g_union;
    const float g_area = g_union;
    const float g_iw = (iw_raw >= 0.f) ? g_inter * ih : 0.f, g_ih = (ih_raw >= 0.f) ? g_inter * iw : 0.f;
    const float g_ew = (ew_raw >= 0.f) ? g_hull * eh : 0.f, g_eh = (eh_raw >= 0.f) ? g_hull * ew : 0.f;
    float gx0 = 0.f, gx1 = 0.f, gy0 = 0.f, gy1 = 0.f;
    gx1 += (ax1 < bx1) ? g_iw : 0.f;  gx0 -= (ax0 > bx0) ? g_iw : 0.f;
    gy1 += (ay1 < by1) ? g_ih : 0.f;  gy0 -= (ay0 > by0) ? g_ih : 0.f;
    gx1 += (ax1 > bx1) ? g_ew : 0.f;  gx0 -= (ax0 < bx0) ? g_ew : 0.f;
    gy1 += (ay1 > by1) ? g_eh : 0.f;  gy0 -= (ay0 < by0) ? g_eh : 0.f;
    gx1 += g_area * ah; gx0 -= g_area * ah;
    gy1 += g_area * aw; gy0 -= g_area * aw;
    r.g[0] = gx0 + gx1; r.g[1] = gy0 + gy1;
    r.g[2] = 0.5f * (gx1 - gx0); r.g[3] = 0.5f * (gy1 - gy0);
    return r;
}

// MODE 0: forward (losses). MODE 1: backward (dlogits, dboxes from the upstream gradients of the losses).
template <int MODE>
__global__ __launch_bounds__(CRIT_THREADS) void criterion_kernel(
    const float* __restrict__ logits, const float* __restrict__ boxes, const float* __restrict__ tgt_box,
    const float* __restrict__ pos_map, const int* __restrict__ tgt_off, const int* __restrict__ match_off,
    const long long* __restrict__ src_idx, const long long* __restrict__ tgt_idx, const float* __restrict__ num_boxes,
    int L, int B, int Q, int K, float eos_coef,
    float* __restrict__ losses,             // MODE 0: [L,4] (+=) : ce, bbox, giou, cardinality
    const float* __restrict__ upstream,     // MODE 1: [L,4] gradient of the total w.r.t. each loss
    float* __restrict__ dlogits, float* __restrict__ dboxes,
    const int* __restrict__ match_status) { // MODE 0: optional [L*B]; non-zero = invalid cost block -> NaN losses
    extern __shared__ int qmap[];           // [Q] global target row matched to query q, or -1
    __shared__ float red[CRIT_THREADS / 64];
    const int lb = blockIdx.x, l = lb / B, b = lb % B;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int Mtot = match_off[B];
    const int m_beg = match_off[b], m_end = match_off[b + 1];
    const int T = tgt_off[b + 1] - tgt_off[b];
    for (int q = tid; q < Q; q += CRIT_THREADS) qmap[q] = -1;
    __syncthreads();
    for (int m = m_beg + tid; m < m_end; m += CRIT_THREADS)
        qmap[(int)src_idx[(size_t)l * Mtot + m]] = tgt_off[b] + (int)tgt_idx[(size_t)l * Mtot + m];
    __syncthreads();

    const float inv_nb = 1.f / num_boxes[0];
    const float* lg = logits + ((size_t)(l * B + b) * Q) * K;
    const float* bx = boxes + ((size_t)(l * B + b) * Q) * 4;
    float g_ce = 0.f, g_l1 = 0.f, g_gi = 0.f;
    if (MODE == 1) {
        g_ce = upstream[l * 4 + 0] * inv_nb;
        g_l1 = upstream[l * 4 + 1] * inv_nb;
        g_gi = upstream[l * 4 + 2] * inv_nb;
    }
    float ce_acc = 0.f, l1_acc = 0.f, gi_acc = 0.f, card = 0.f;
    for (int q = wave; q < Q; q += CRIT_THREADS / 64) {
        const float* row = lg + (size_t)q * K;
        float x[8];
        float mx = -INFINITY;
        int arg = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int k = lane + 64 * i;
            x[i] = (k < K) ? row[k] : -INFINITY;
            if (x[i] > mx) { mx = x[i]; arg = k; }
        }
        // wave arg-max (first index wins on ties, like torch.argmax on the CPU/GPU for distinct values)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float om = __shfl_xor(mx, o, 64);
            const int oa = __shfl_xor(arg, o, 64);
            if (om > mx || (om == mx && oa < arg)) { mx = om; arg = oa; }
        }
        float se = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) se += (lane + 64 * i < K) ? __expf(x[i] - mx) : 0.f;
        se = wave_sum(se);
        const float lse = mx + __logf(se);
        const int t = qmap[q];
        if (t >= 0) {
            const float* pm = pos_map + (size_t)t * K;
            float dot = 0.f, tsum = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int k = lane + 64 * i;
                if (k < K) { dot += (x[i] - lse) * pm[k]; tsum += pm[k]; }
            }
            if (MODE == 0) {
                dot = wave_sum(dot);
                if (lane == 0) ce_acc += -dot;
            } else {
                tsum = wave_sum(tsum);
                float* dl = dlogits + ((size_t)(l * B + b) * Q + q) * K;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int k = lane + 64 * i;
                    if (k < K) dl[k] = g_ce * (__expf(x[i] - lse) * tsum - pm[k]);
                }
            }
            if (lane == 0) {
                const float* pb = bx + q * 4;
                const float* tb = tgt_box + (size_t)t * 4;
                const GiouPair gp = giou_pair(pb, tb);
                if (MODE == 0) {
                    l1_acc += ((fabsf(pb[0] - tb[0]) + fabsf(pb[1] - tb[1])) + fabsf(pb[2] - tb[2])) + fabsf(pb[3] - tb[3]);
                    gi_acc += 1.f - gp.giou;
                } else {
                    float* db = dboxes + ((size_t)(l * B + b) * Q + q) * 4;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float d = pb[c] - tb[c];
                        const float sgn = (d > 0.f) ? 1.f : ((d < 0.f) ? -1.f : 0.f);
                        db[c] = g_l1 * sgn - g_gi * gp.g[c];
                    }
                }
            }
        } else {
            // unmatched: one-hot target on the no-object slot (last class), weight eos_coef
            if (MODE == 0) {
                if (lane == ((K - 1) & 63)) ce_acc += -(x[(K - 1) >> 6] - lse) * eos_coef;
            } else {
                float* dl = dlogits + ((size_t)(l * B + b) * Q + q) * K;
                const float ge = g_ce * eos_coef;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int k = lane + 64 * i;
                    if (k < K) dl[k] = ge * (__expf(x[i] - lse) - (k == K - 1 ? 1.f : 0.f));
                }
                if (lane < 4) dboxes[((size_t)(l * B + b) * Q + q) * 4 + lane] = 0.f;
            }
        }
        if (MODE == 0 && lane == 0 && arg != K - 1) card += 1.f;
    }
    if (MODE == 0) {
        const float ce = block_sum(ce_acc, red);
        const float l1 = block_sum(l1_acc, red);
        const float gi = block_sum(gi_acc, red);
        const float cd = block_sum(card, red);
        if (tid == 0) {
            if (match_status != nullptr && match_status[lb] != 0) {   // SciPy raises here (matcher.py:85): poison instead of a host sync
                const float nan = __int_as_float(0x7fc00000);
                for (int c = 0; c < 4; ++c) atomicAdd(losses + l * 4 + c, nan);
            }
            atomicAdd(losses + l * 4 + 0, ce * inv_nb);
            atomicAdd(losses + l * 4 + 1, l1 * inv_nb);
            atomicAdd(losses + l * 4 + 2, gi * inv_nb);
            atomicAdd(losses + l * 4 + 3, fabsf(cd - (float)T) / (float)B);
        }
    }
}

}  // namespace toist

using namespace toist;

static int criterion_args_ok(int L, int B, int Q, int K) {
    TOIST_REQUIRE(L > 0 && B > 0 && Q > 0 && K > 0 && K <= 512, "toist_criterion: bad shape L=%d B=%d Q=%d K=%d (K<=512)", L, B, Q, K);
    return TOIST_OK;
}

extern "C" int toist_criterion_fwd(const float* logits, const float* boxes, const float* tgt_boxes, const float* pos_map,
                                   const int32_t* tgt_off, const int32_t* match_off, const int64_t* src_idx, const int64_t* tgt_idx,
                                   const float* num_boxes, int L, int B, int Q, int K, float eos_coef, float* losses,
                                   const int32_t* match_status, void* stream) {
    if (int rc = criterion_args_ok(L, B, Q, K)) return rc;
    hipLaunchKernelGGL(criterion_kernel<0>, dim3(L * B), dim3(CRIT_THREADS), sizeof(int) * Q, (hipStream_t)stream, logits, boxes, tgt_boxes, pos_map,
                       tgt_off, match_off, (const long long*)src_idx, (const long long*)tgt_idx, num_boxes, L, B, Q, K, eos_coef, losses,
                       (const float*)nullptr, (float*)nullptr, (float*)nullptr, (const int*)match_status);
    return check_launch("toist_criterion_fwd");
}

extern "C" int toist_criterion_bwd(const float* logits, const float* boxes, const float* tgt_boxes, const float* pos_map,
                                   const int32_t* tgt_off, const int32_t* match_off, const int64_t* src_idx, const int64_t* tgt_idx,
                                   const float* num_boxes, int L, int B, int Q, int K, float eos_coef, const float* upstream,
                                   float* dlogits, float* dboxes, void* stream) {
    if (int rc = criterion_args_ok(L, B, Q, K)) return rc;
    hipLaunchKernelGGL(criterion_kernel<1>, dim3(L * B), dim3(CRIT_THREADS), sizeof(int) * Q, (hipStream_t)stream, logits, boxes, tgt_boxes, pos_map,
                       tgt_off, match_off, (const long long*)src_idx, (const long long*)tgt_idx, num_boxes, L, B, Q, K, eos_coef,
                       (float*)nullptr, upstream, dlogits, dboxes, (const int*)nullptr);
    return check_launch("toist_criterion_bwd");
}
