// Evaluation-side mask kernels: the work PostProcessSegm + pycocotools do between the mask head and COCOeval
// (reference: models/postprocessors.py:73-109, datasets/coco_eval.py:307-332 -> mask_util.encode / maskUtils.iou).
//
// The reference materialises two fp32 resizes of [B, Q, H, W], copies Q dense masks per image to the host and run-length
// encodes / intersects them on one core.  Here a mask lives in HBM as a COLUMN-MAJOR BIT PLANE
//     bits[mask][x][yw]   (uint64; bit b of word yw is pixel (y = 64*yw + b, x); bits beyond H are 0)
// which is 1/32 of an fp32 mask and already in RLE's pixel order (pixel index = x*H + y):
//   * mask_resize_pack : both bilinear resizes + sigmoid + threshold fused, straight from the [h0, w0] mask logits
//   * mask_pack/unpack : dense bool <-> bit plane (ground truth in, reference-format results out)
//   * mask_area / mask_iou : popcounts; IoU = i / (crowd ? area_d : area_d + area_g - i) in double, as rleIou does
//   * mask_rle_count / emit / counts : run lengths (zeros first) = differences of the transition positions
// All of it is HBM/L2-bound bit work: one wave owns a 64 x 64 pixel tile, lanes along x (coalesced sources), each lane
// building the 64-bit word of its own column.
#include "common.h"

namespace toist {

static constexpr int EM_THREADS = 256;

__device__ __forceinline__ int words_of(int h) { return (h + 63) >> 6; }

// torch's upsample_bilinear2d source index (align_corners = false): scale * (dst + 0.5) - 0.5, clamped at 0
struct Tap {
    int i0, i1;
    float l0, l1;
};
__device__ __forceinline__ Tap tap_of(int dst, float scale, int in_size) {
    float s = scale * ((float)dst + 0.5f) - 0.5f;
    s = s < 0.f ? 0.f : s;
    Tap t;
    t.i0 = (int)s;
    if (t.i0 > in_size - 1) t.i0 = in_size - 1;
    t.i1 = t.i0 + (t.i0 < in_size - 1 ? 1 : 0);
    t.l1 = s - (float)t.i0;
    t.l0 = 1.f - t.l1;
    return t;
}

__device__ __forceinline__ float bilerp(const float* __restrict__ p, int ld, const Tap& ty, const Tap& tx) {
    return ty.l0 * (tx.l0 * p[(size_t)ty.i0 * ld + tx.i0] + tx.l1 * p[(size_t)ty.i0 * ld + tx.i1]) +
           ty.l1 * (tx.l0 * p[(size_t)ty.i1 * ld + tx.i0] + tx.l1 * p[(size_t)ty.i1 * ld + tx.i1]);
}

// grid: (ceil(W/64) * YW / 4, n_masks); wave = one (x tile, y word)
__global__ __launch_bounds__(EM_THREADS) void mask_resize_pack_kernel(const float* __restrict__ src, int h0, int w0, int max_h, int max_w, int crop_h,
                                                                       int crop_w, int H, int W, float threshold, uint64_t* __restrict__ bits) {
    const int yw_n = words_of(H), xt_n = (W + 63) >> 6;
    const int unit = blockIdx.x * (EM_THREADS / 64) + (threadIdx.x >> 6);
    if (unit >= yw_n * xt_n) return;
    const int yw = unit % yw_n, x = (unit / yw_n) * 64 + (threadIdx.x & 63);
    const int m = blockIdx.y;
    const float* p = src + (size_t)m * h0 * w0;
    // stage 1: [h0, w0] -> [max_h, max_w]; stage 2: its [crop_h, crop_w] corner -> [H, W]
    const float s1y = (float)h0 / (float)max_h, s1x = (float)w0 / (float)max_w;
    const float s2y = (float)crop_h / (float)H, s2x = (float)crop_w / (float)W;
    if (x >= W) return;
    const Tap bx = tap_of(x, s2x, crop_w);
    const Tap ax0 = tap_of(bx.i0, s1x, w0), ax1 = tap_of(bx.i1, s1x, w0);
    uint64_t word = 0;
    const int y_end = min(64, H - yw * 64);
    for (int b = 0; b < y_end; ++b) {
        const Tap by = tap_of(yw * 64 + b, s2y, crop_h);
        const Tap ay0 = tap_of(by.i0, s1y, h0), ay1 = tap_of(by.i1, s1y, h0);
        const float v00 = bilerp(p, w0, ay0, ax0), v01 = bilerp(p, w0, ay0, ax1);
        const float v10 = bilerp(p, w0, ay1, ax0), v11 = bilerp(p, w0, ay1, ax1);
        const float v = by.l0 * (bx.l0 * v00 + bx.l1 * v01) + by.l1 * (bx.l0 * v10 + bx.l1 * v11);
        const float prob = 1.f / (1.f + expf(-v));
        word |= (uint64_t)(prob > threshold) << b;
    }
    bits[((size_t)m * W + x) * yw_n + yw] = word;
}

__global__ __launch_bounds__(EM_THREADS) void mask_pack_kernel(const uint8_t* __restrict__ dense, int H, int W, uint64_t* __restrict__ bits) {
    const int yw_n = words_of(H), xt_n = (W + 63) >> 6;
    const int unit = blockIdx.x * (EM_THREADS / 64) + (threadIdx.x >> 6);
    if (unit >= yw_n * xt_n) return;
    const int yw = unit % yw_n, x = (unit / yw_n) * 64 + (threadIdx.x & 63);
    if (x >= W) return;
    const uint8_t* p = dense + (size_t)blockIdx.y * H * W;
    uint64_t word = 0;
    const int y_end = min(64, H - yw * 64);
    for (int b = 0; b < y_end; ++b) word |= (uint64_t)(p[(size_t)(yw * 64 + b) * W + x] != 0) << b;
    bits[((size_t)blockIdx.y * W + x) * yw_n + yw] = word;
}

__global__ __launch_bounds__(EM_THREADS) void mask_unpack_kernel(const uint64_t* __restrict__ bits, int H, int W, uint8_t* __restrict__ dense) {
    const int yw_n = words_of(H), xt_n = (W + 63) >> 6;
    const int unit = blockIdx.x * (EM_THREADS / 64) + (threadIdx.x >> 6);
    if (unit >= yw_n * xt_n) return;
    const int yw = unit % yw_n, x = (unit / yw_n) * 64 + (threadIdx.x & 63);
    if (x >= W) return;
    uint8_t* p = dense + (size_t)blockIdx.y * H * W;
    const uint64_t word = bits[((size_t)blockIdx.y * W + x) * yw_n + yw];
    const int y_end = min(64, H - yw * 64);
    for (int b = 0; b < y_end; ++b) p[(size_t)(yw * 64 + b) * W + x] = (uint8_t)((word >> b) & 1);
}

__device__ __forceinline__ unsigned block_sum_u32(unsigned v) {
    __shared__ unsigned part[EM_THREADS / 64];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
    __syncthreads();
    unsigned t = 0;
#pragma unroll
    for (int i = 0; i < EM_THREADS / 64; ++i) t += part[i];
    return t;
}

__global__ __launch_bounds__(EM_THREADS) void mask_area_kernel(const uint64_t* __restrict__ bits, size_t words, uint32_t* __restrict__ area) {
    const uint64_t* p = bits + (size_t)blockIdx.x * words;
    unsigned acc = 0;
    for (size_t i = threadIdx.x; i < words; i += EM_THREADS) acc += __popcll(p[i]);
    acc = block_sum_u32(acc);
    if (threadIdx.x == 0) area[blockIdx.x] = acc;
}

// grid (n_gt, n_dt): iou[d, g]
__global__ __launch_bounds__(EM_THREADS) void mask_iou_kernel(const uint64_t* __restrict__ dt, const uint64_t* __restrict__ gt, size_t words,
                                                               const uint8_t* __restrict__ iscrowd, const uint32_t* __restrict__ area_d,
                                                               const uint32_t* __restrict__ area_g, int n_gt, double* __restrict__ iou) {
    const int g = blockIdx.x, d = blockIdx.y;
    const uint64_t *a = dt + (size_t)d * words, *b = gt + (size_t)g * words;
    unsigned acc = 0;
    for (size_t i = threadIdx.x; i < words; i += EM_THREADS) acc += __popcll(a[i] & b[i]);
    acc = block_sum_u32(acc);
    if (threadIdx.x == 0) {
        double r = 0.0;
        if (acc) {
            const double u = iscrowd[g] ? (double)area_d[d] : (double)area_d[d] + (double)area_g[g] - (double)acc;
            r = (double)acc / u;
        }
        iou[(size_t)d * n_gt + g] = r;
    }
}

// value of the pixel that precedes column x in RLE order (0 before the first pixel)
__device__ __forceinline__ uint64_t carry_in(const uint64_t* __restrict__ mask, int x, int H) {
    if (x == 0) return 0;
    const int yw_n = words_of(H);
    return (mask[(size_t)(x - 1) * yw_n + ((H - 1) >> 6)] >> ((H - 1) & 63)) & 1;
}

// one thread per (mask, column): number of value changes inside the column (incl. against the previous column's last pixel)
__global__ __launch_bounds__(EM_THREADS) void mask_rle_count_kernel(const uint64_t* __restrict__ bits, int n, int H, int W, int32_t* __restrict__ count) {
    const size_t idx = (size_t)blockIdx.x * EM_THREADS + threadIdx.x;
    if (idx >= (size_t)n * W) return;
    const int m = (int)(idx / W), x = (int)(idx % W), yw_n = words_of(H);
    const uint64_t* mask = bits + (size_t)m * W * yw_n;
    uint64_t carry = carry_in(mask, x, H);
    int c = 0;
    for (int yw = 0; yw < yw_n; ++yw) {
        const uint64_t w = mask[(size_t)x * yw_n + yw];
        uint64_t t = w ^ ((w << 1) | carry);
        const int valid = min(64, H - yw * 64);
        if (valid < 64) t &= (1ull << valid) - 1;
        c += __popcll(t);
        carry = (w >> (valid - 1)) & 1;
    }
    count[idx] = c;
}

// offset[m, x] = index (in `pos`) of the column's first transition; positions are pixel indices x*H + y
__global__ __launch_bounds__(EM_THREADS) void mask_rle_emit_kernel(const uint64_t* __restrict__ bits, int n, int H, int W,
                                                                    const int64_t* __restrict__ offset, uint32_t* __restrict__ pos) {
    const size_t idx = (size_t)blockIdx.x * EM_THREADS + threadIdx.x;
    if (idx >= (size_t)n * W) return;
    const int m = (int)(idx / W), x = (int)(idx % W), yw_n = words_of(H);
    const uint64_t* mask = bits + (size_t)m * W * yw_n;
    uint64_t carry = carry_in(mask, x, H);
    uint32_t* out = pos + offset[idx];
    for (int yw = 0; yw < yw_n; ++yw) {
        const uint64_t w = mask[(size_t)x * yw_n + yw];
        uint64_t t = w ^ ((w << 1) | carry);
        const int valid = min(64, H - yw * 64);
        if (valid < 64) t &= (1ull << valid) - 1;
        while (t) {
            const int b = __ffsll((long long)t) - 1;
            *out++ = (uint32_t)x * (uint32_t)H + (uint32_t)(yw * 64 + b);
            t &= t - 1;
        }
        carry = (w >> (valid - 1)) & 1;
    }
}

// run r of mask m: first[m] + r ... ; runs[m] = transitions + 1; counts = differences of 0, pos..., H*W
__global__ __launch_bounds__(EM_THREADS) void mask_rle_counts_kernel(const uint32_t* __restrict__ pos, const int64_t* __restrict__ first_pos,
                                                                      const int64_t* __restrict__ first_run, int n, uint32_t hw,
                                                                      uint32_t* __restrict__ counts) {
    const int m = blockIdx.y;
    const int64_t n_tr = first_pos[m + 1] - first_pos[m];
    const uint32_t* p = pos + first_pos[m];
    uint32_t* c = counts + first_run[m];
    for (int64_t r = (int64_t)blockIdx.x * EM_THREADS + threadIdx.x; r <= n_tr; r += (int64_t)gridDim.x * EM_THREADS) {
        const uint32_t lo = r == 0 ? 0u : p[r - 1], hi = r == n_tr ? hw : p[r];
        c[r] = hi - lo;
    }
}


// COCOeval.evaluateImg for a batch of images (pycocotools cocoeval.py, called from datasets/coco_eval.py:368-399): one thread
// per (image, area range, IoU threshold) walks the score-sorted detections and greedily takes, for each, the best still-free
// ground truth (crowd ground truth may be taken repeatedly; a non-ignored match is never traded for an ignored one).  The
// reference sorts the ground truth "non-ignored first" (stable): two passes over the original order visit it identically.
__global__ __launch_bounds__(64) void coco_match_kernel(const double* __restrict__ iou, const int64_t* __restrict__ iou_off,
                                                         const double* __restrict__ dt_area, const int64_t* __restrict__ dt_off,
                                                         const double* __restrict__ gt_area, const uint8_t* __restrict__ gt_ignore,
                                                         const uint8_t* __restrict__ gt_crowd, const int64_t* __restrict__ gt_off,
                                                         const double* __restrict__ area_rng, int A, const double* __restrict__ thrs, int T,
                                                         int32_t* __restrict__ dt_match, uint8_t* __restrict__ dt_ignore,
                                                         uint8_t* __restrict__ gt_range_ignore, uint8_t* __restrict__ gt_taken) {
    const int img = blockIdx.x;
    const int D = (int)(dt_off[img + 1] - dt_off[img]), G = (int)(gt_off[img + 1] - gt_off[img]);
    const double* io = iou + iou_off[img];
    const double* da = dt_area + dt_off[img];
    const double* ga = gt_area + gt_off[img];
    const uint8_t *gi = gt_ignore + gt_off[img], *gc = gt_crowd + gt_off[img];
    for (int at = threadIdx.x; at < A * T; at += 64) {
        const int a = at / T, t = at % T;
        const double lo = area_rng[2 * a], hi = area_rng[2 * a + 1];
        int32_t* match = dt_match + (size_t)A * T * dt_off[img] + (size_t)at * D;
        uint8_t* ign = dt_ignore + (size_t)A * T * dt_off[img] + (size_t)at * D;
        uint8_t* taken = gt_taken + (size_t)A * T * gt_off[img] + (size_t)at * G;
        uint8_t* gflag = gt_range_ignore + (size_t)A * gt_off[img] + (size_t)a * G;      // same bytes from every t of this a
        for (int g = 0; g < G; ++g) {
            taken[g] = 0;
            gflag[g] = (gi[g] || ga[g] < lo || ga[g] > hi) ? 1 : 0;
        }
        for (int d = 0; d < D; ++d) {
            double best = thrs[t] < 1.0 - 1e-10 ? thrs[t] : 1.0 - 1e-10;
            int m = -1, m_flag = 0;
            bool stop = false;
            for (int pass = 0; pass < 2 && !stop; ++pass)
                for (int g = 0; g < G; ++g) {
                    const int flag = (gi[g] || ga[g] < lo || ga[g] > hi) ? 1 : 0;
                    if (flag != pass) continue;
                    if (taken[g] && !gc[g]) continue;
                    if (m > -1 && m_flag == 0 && flag == 1) {
                        stop = true;
                        break;
                    }
                    const double v = io[(size_t)d * G + g];
                    if (v < best) continue;
                    best = v;
                    m = g;
                    m_flag = flag;
                }
            match[d] = m;
            if (m >= 0) {
                taken[m] = 1;
                ign[d] = (uint8_t)m_flag;
            } else {
                ign[d] = (da[d] < lo || da[d] > hi) ? 1 : 0;
            }
        }
    }
}

static inline dim3 tile_grid(int n, int H, int W) {
    const int units = ((H + 63) / 64) * ((W + 63) / 64);
    return dim3((units + EM_THREADS / 64 - 1) / (EM_THREADS / 64), n);
}

}  // namespace toist

using namespace toist;

#define EM_GEOMETRY(name)                                                                                        \
    TOIST_REQUIRE(n >= 0 && h > 0 && w > 0 && (long long)h * w < (1ll << 32) && n <= 65535, name ": bad geometry"); \
    if (n == 0) return TOIST_OK

extern "C" int toist_mask_resize_pack(const float* src, int n, int h0, int w0, int max_h, int max_w, int crop_h, int crop_w, int h, int w,
                                      float threshold, uint64_t* bits, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    EM_GEOMETRY("toist_mask_resize_pack");
    TOIST_REQUIRE(src && bits, "toist_mask_resize_pack: null pointer");
    TOIST_REQUIRE(h0 > 0 && w0 > 0 && crop_h > 0 && crop_w > 0 && crop_h <= max_h && crop_w <= max_w,
                  "toist_mask_resize_pack: the crop [%d, %d] must lie inside the padded size [%d, %d]", crop_h, crop_w, max_h, max_w);
    hipLaunchKernelGGL(mask_resize_pack_kernel, tile_grid(n, h, w), dim3(EM_THREADS), 0, stream, src, h0, w0, max_h, max_w, crop_h, crop_w, h, w,
                       threshold, bits);
    return check_launch("toist_mask_resize_pack");
}

extern "C" int toist_mask_pack(const uint8_t* dense, int n, int h, int w, uint64_t* bits, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    EM_GEOMETRY("toist_mask_pack");
    TOIST_REQUIRE(dense && bits, "toist_mask_pack: null pointer");
    hipLaunchKernelGGL(mask_pack_kernel, tile_grid(n, h, w), dim3(EM_THREADS), 0, stream, dense, h, w, bits);
    return check_launch("toist_mask_pack");
}

extern "C" int toist_mask_unpack(const uint64_t* bits, int n, int h, int w, uint8_t* dense, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    EM_GEOMETRY("toist_mask_unpack");
    TOIST_REQUIRE(dense && bits, "toist_mask_unpack: null pointer");
    hipLaunchKernelGGL(mask_unpack_kernel, tile_grid(n, h, w), dim3(EM_THREADS), 0, stream, bits, h, w, dense);
    return check_launch("toist_mask_unpack");
}

extern "C" int toist_mask_area(const uint64_t* bits, int n, int h, int w, uint32_t* area, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    EM_GEOMETRY("toist_mask_area");
    TOIST_REQUIRE(bits && area, "toist_mask_area: null pointer");
    hipLaunchKernelGGL(mask_area_kernel, dim3(n), dim3(EM_THREADS), 0, stream, bits, (size_t)w * ((h + 63) / 64), area);
    return check_launch("toist_mask_area");
}

extern "C" int toist_mask_iou(const uint64_t* dt, int n_dt, const uint64_t* gt, int n_gt, const uint8_t* iscrowd, const uint32_t* area_dt,
                              const uint32_t* area_gt, int h, int w, double* iou, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    TOIST_REQUIRE(n_dt >= 0 && n_gt >= 0 && n_dt <= 65535 && h > 0 && w > 0, "toist_mask_iou: bad geometry");
    if (n_dt == 0 || n_gt == 0) return TOIST_OK;
    TOIST_REQUIRE(dt && gt && iscrowd && area_dt && area_gt && iou, "toist_mask_iou: null pointer");
    hipLaunchKernelGGL(mask_iou_kernel, dim3(n_gt, n_dt), dim3(EM_THREADS), 0, stream, dt, gt, (size_t)w * ((h + 63) / 64), iscrowd, area_dt,
                       area_gt, n_gt, iou);
    return check_launch("toist_mask_iou");
}

extern "C" int toist_mask_rle_count(const uint64_t* bits, int n, int h, int w, int32_t* column_transitions, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    EM_GEOMETRY("toist_mask_rle_count");
    TOIST_REQUIRE(bits && column_transitions, "toist_mask_rle_count: null pointer");
    const size_t cols = (size_t)n * w;
    hipLaunchKernelGGL(mask_rle_count_kernel, dim3((unsigned)((cols + EM_THREADS - 1) / EM_THREADS)), dim3(EM_THREADS), 0, stream, bits, n, h, w,
                       column_transitions);
    return check_launch("toist_mask_rle_count");
}

extern "C" int toist_mask_rle_emit(const uint64_t* bits, int n, int h, int w, const int64_t* column_offset, uint32_t* positions, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    EM_GEOMETRY("toist_mask_rle_emit");
    TOIST_REQUIRE(bits && column_offset && positions, "toist_mask_rle_emit: null pointer");
    const size_t cols = (size_t)n * w;
    hipLaunchKernelGGL(mask_rle_emit_kernel, dim3((unsigned)((cols + EM_THREADS - 1) / EM_THREADS)), dim3(EM_THREADS), 0, stream, bits, n, h, w,
                       column_offset, positions);
    return check_launch("toist_mask_rle_emit");
}

extern "C" int toist_mask_rle_counts(const uint32_t* positions, const int64_t* first_position, const int64_t* first_run, int n, int h, int w,
                                     uint32_t* counts, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    EM_GEOMETRY("toist_mask_rle_counts");
    TOIST_REQUIRE(first_position && first_run && counts, "toist_mask_rle_counts: null pointer");
    hipLaunchKernelGGL(mask_rle_counts_kernel, dim3(32, n), dim3(EM_THREADS), 0, stream, positions, first_position, first_run, n,
                       (uint32_t)((long long)h * w), counts);
    return check_launch("toist_mask_rle_counts");
}

extern "C" int toist_coco_match(const double* iou, const int64_t* iou_offset, const double* dt_area, const int64_t* dt_offset, const double* gt_area,
                                const uint8_t* gt_ignore, const uint8_t* gt_crowd, const int64_t* gt_offset, int n_images, const double* area_ranges,
                                int n_ranges, const double* iou_thresholds, int n_thresholds, int32_t* dt_match, uint8_t* dt_ignore,
                                uint8_t* gt_range_ignore, uint8_t* gt_taken, void* stream) {
    TOIST_REQUIRE(n_images >= 0 && n_ranges > 0 && n_thresholds > 0, "toist_coco_match: bad extents");
    if (n_images == 0) return TOIST_OK;
    TOIST_REQUIRE(iou_offset && dt_offset && gt_offset && area_ranges && iou_thresholds, "toist_coco_match: null table");
    hipLaunchKernelGGL(coco_match_kernel, dim3(n_images), dim3(64), 0, (hipStream_t)stream, iou, iou_offset, dt_area, dt_offset, gt_area, gt_ignore,
                       gt_crowd, gt_offset, area_ranges, n_ranges, iou_thresholds, n_thresholds, dt_match, dt_ignore, gt_range_ignore, gt_taken);
    return check_launch("toist_coco_match");
}
