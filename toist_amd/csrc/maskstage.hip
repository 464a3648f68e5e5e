// One pass per stage of the mask head's FPN tail (/root/reference/models/segmentation.py:203-241, MaskHeadSmallConv.forward):
//     x = adapter(fpn) + interpolate(x, nearest 2x);  x = relu(gn(lay(x)))            (lay3..lay5)      and      x = out_lay(x)
// Per-op launches write and re-read every intermediate of the 800 maps: the upsampled sum (1.3 GB at 160 x 160), the convolution output, a
// statistics pass, the normalised activation (0.66 GB), an 8-channel padded out_lay output.  Here a workgroup owns a strip of 16 x 16 output
// tiles of one map and builds the convolution's INPUT in LDS from what the previous stage left in HBM:
//   GNIN : the source is the previous convolution's raw output; GroupNorm(8, CIN) + ReLU are applied on the way in, from that stage's
//          {sum, sum of squares} per (map, group) -- the normalised activation is never written;
//   UP   : the convolution is linear, so lay(adapter(fpn) + up2(x)) = lay(adapter(fpn)) + lay_nobias(up2(x)): the first term is ONE small
//          convolution per IMAGE (the caller's `fpn_conv` [N/Q,H,W,COUT], bias included) added in the epilogue, and the second reads the
//          upsampled activation straight from the 10 x 10 SOURCE pixels under the tile (tap (dy,dx) of output pixel (y,x) is source pixel
//          ((y+dy-1)>>1, (x+dx-1)>>1)) -- neither the upsampled sum nor the FPN add per upsampled pixel exists any more;
// then runs the 3x3 as nine taps of MFMAs out of LDS (weights in registers, D^T = W.X^T: a lane ends up with 4 consecutive output channels of
// one pixel), writes the raw convolution output once and folds its GroupNorm statistics in the epilogue (OUT1: the single out_lay channel
// goes out as f32 [N,H,W], no statistics).  HBM traffic per stage: one read of the (4x smaller) source + one write of the output.
#include "common.h"

namespace toist {

typedef __attribute__((ext_vector_type(4))) short ms_bf16x4_t;

template <int KM>
struct MsFrag {
    typedef typename std::conditional<KM == 32, bf16x8_t, ms_bf16x4_t>::type type;
};
template <int KM>
__device__ __forceinline__ f32x4_t ms_mfma(typename MsFrag<KM>::type b, typename MsFrag<KM>::type a, f32x4_t c) {
    if constexpr (KM == 32) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a, c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(b, a, c, 0, 0, 0);
}

// LDS tile: pixel-major, CIN channels per pixel (16-byte chunks) + one chunk of padding when a pixel is 64 bytes or more: a 16-lane group of a
// fragment read takes the same chunk of 16 consecutive pixels, and with the pixel stride at 80 / 144 bytes those reads start in different
// 4-bank groups (32-byte pixels with 8-byte reads are contiguous as they are).  No swizzle: every tap is a compile-time offset from a per-lane base.
typedef __attribute__((ext_vector_type(2))) float ms_f2;

__device__ __forceinline__ void ms_unpack8v(const uint4& u, ms_f2* f) {
    f[0] = ms_f2{__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u)};
    f[1] = ms_f2{__uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u)};
    f[2] = ms_f2{__uint_as_float(u.z << 16), __uint_as_float(u.z & 0xffff0000u)};
    f[3] = ms_f2{__uint_as_float(u.w << 16), __uint_as_float(u.w & 0xffff0000u)};
}
__device__ __forceinline__ uint4 ms_pack8v(const ms_f2* f) {
    return make_uint4(pack2bf(f[0].x, f[0].y), pack2bf(f[1].x, f[1].y), pack2bf(f[2].x, f[2].y), pack2bf(f[3].x, f[3].y));
}

template <int CIN, int COUT, bool GNIN, bool UP, bool OUT1>
__global__ __launch_bounds__(256, (CIN >= 64 ? 2 : 3)) void mask_stage_kernel(const bf16_t* __restrict__ src, const float* __restrict__ src_stats, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, const bf16_t* __restrict__ fpn_conv, const bf16_t* __restrict__ w,
                                                          const float* __restrict__ bias, bf16_t* __restrict__ out, float* __restrict__ out_stats,
                                                          float* __restrict__ out1, int Q, int H, int W, int w_rows, float eps) {
    constexpr int CH = CIN / 8;
    constexpr int TS = 16, HS = TS + 2;
    constexpr int SP = UP ? TS / 2 + 2 : HS;            // side of the LDS tile in source pixels: 10 x 10 under an upsampled tile, else the 18 x 18 halo tile
    constexpr int PX = (CH >= 4) ? CIN + 8 : CIN;       // elements per pixel of the LDS tile (one chunk of padding)
    constexpr int KM = (CIN >= 32) ? 32 : 16;
    constexpr int KS = CIN / KM;
    constexpr int KL = KM / 4;
    constexpr int NB = (COUT + 15) / 16;
    constexpr int CG = OUT1 ? 1 : COUT / 8;          // channels per output GroupNorm group
    static_assert(OUT1 || CG == 2 || CG == 4 || CG == 8, "output groups of 2, 4 or 8 channels");
    static_assert(256 % CH == 0, "a thread keeps one channel chunk");
    typedef typename MsFrag<KM>::type frag_t;
    __shared__ __attribute__((aligned(16))) bf16_t tile[2][SP * SP * PX];     // two buffers: tile k+1 is written while slower waves still read tile k -> ONE barrier per tile
    __shared__ float sacc[2][8];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c16 = lane & 15;
    const int tiles_y = (H + TS - 1) / TS, tiles_x = (W + TS - 1) / TS;
    const int n = blockIdx.x / tiles_y, ty = blockIdx.x - n * tiles_y;
    const int y0 = ty * TS;
    const int SH = UP ? (H >> 1) : H, SW = UP ? (W >> 1) : W;
    const int cc = tid % CH;                          // this thread's channel chunk in the fill phase

    // ---- weights -> registers: row (nb*16 + c16) of w[row][tap][CIN], channels ks*KM + g*KL .. +KL-1 ----
    frag_t bw[9][NB][KS];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int row = nb * 16 + c16;
                union { uint4 q; uint2 d; frag_t v; } f;
                f.q = make_uint4(0, 0, 0, 0);
                if (row < w_rows) {
                    const bf16_t* p = w + ((size_t)row * 9 + t) * CIN + ks * KM + g * KL;
                    if constexpr (KL == 8) f.q = *reinterpret_cast<const uint4*>(p);
                    else f.d = *reinterpret_cast<const uint2*>(p);
                }
                bw[t][nb][ks] = f.v;
            }
    float bs[NB][4];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = nb * 16 + g * 4 + j;
            bs[nb][j] = (bias != nullptr && c < (OUT1 ? 1 : COUT)) ? bias[c] : 0.f;
        }
    // ---- GroupNorm + ReLU coefficients of this thread's 8 source channels: y = max(x * ca + cb, 0) (the arithmetic of gn_apply_kernel) ----
    ms_f2 ca[4], cb[4];
    if constexpr (GNIN) {
        constexpr int SCG = CIN / 8;
        const float cnt = (float)SH * (float)SW * (float)SCG;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = cc * 8 + j, gi = c / SCG;
            const float mean = src_stats[((size_t)n * 8 + gi) * 2] / cnt;
            const float var = fmaxf(src_stats[((size_t)n * 8 + gi) * 2 + 1] / cnt - mean * mean, 0.f);
            const float a = rsqrtf(var + eps) * gamma[c];
            ca[j >> 1][j & 1] = a;
            cb[j >> 1][j & 1] = beta[c] - mean * a;
        }
    }
    if (tid < 16) sacc[tid >> 3][tid & 7] = 0.f;
    float ssum[NB][2], ssq[NB][2];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) { ssum[nb][0] = ssum[nb][1] = 0.f; ssq[nb][0] = ssq[nb][1] = 0.f; }

    const bf16_t* src_n = src + (size_t)n * SH * SW * CIN;
    const bf16_t* res_b = UP ? fpn_conv + (size_t)(n / Q) * H * W * COUT : nullptr;

    // ---- fill-phase tasks of this thread, fixed for the whole strip (only the column origin moves): one pixel of the LDS tile x one channel chunk ----
    constexpr int NTASK = SP * SP * CH;
    constexpr int U = (NTASK + 255) / 256;
    int t_i[U], t_j[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int q = tid + 256 * u, sp = q / CH;
        t_i[u] = (q < NTASK) ? sp / SP : -100;           // -100: no task (every bound check below fails)
        t_j[u] = sp - (sp / SP) * SP;
    }
    // per-lane bases of the fragment reads (channel block g).  Halo tile: pixel (row wave*4, column c16).  Source tile: output column c16 reads
    // source column (c16 + dx + 1) >> 1 = jA, jB + 1, jA + 1 for dx = 0, 1, 2 with jA = (c16 + 1) >> 1, jB = c16 >> 1; output row wave*4 + rr reads
    // source row wave*2 + ((rr + dy + 1) >> 1).
    const bf16_t* base_a0 = UP ? tile[0] + (size_t)(wave * 2 * SP + ((c16 + 1) >> 1)) * PX + g * KL : tile[0] + (size_t)(wave * 4 * HS + c16) * PX + g * KL;
    const bf16_t* base_b0 = UP ? tile[0] + (size_t)(wave * 2 * SP + (c16 >> 1)) * PX + g * KL : base_a0;

    // The fill of a tile is split in two: `issue` puts the tile's global loads in flight (raw 16-byte chunks into registers), `commit` normalises
    // and writes the LDS tile.  The loads of tile k+1 are issued before tile k's MFMAs, so a workgroup's strip is not a chain of
    // (load latency -> fill -> barrier -> MFMAs) -- measured per-tile time had been the load latency.
    const int ty0 = UP ? (y0 >> 1) - 1 : y0 - 1;         // tile origin in source coordinates
    uint4 r_src[U];
    auto issue = [&](const int x0, const int u) __attribute__((always_inline)) {
        const int sy = ty0 + t_i[u], sx = (UP ? (x0 >> 1) - 1 : x0 - 1) + t_j[u];
        r_src[u] = make_uint4(0, 0, 0, 0);
        if (t_i[u] >= 0 && (unsigned)sy < (unsigned)SH && (unsigned)sx < (unsigned)SW)
            r_src[u] = *reinterpret_cast<const uint4*>(src_n + ((size_t)sy * SW + sx) * CIN + cc * 8);
    };
    auto commit = [&](const int x0, const int u, bf16_t* buf) __attribute__((always_inline)) {
        if (t_i[u] < 0) return;
        const int sy = ty0 + t_i[u], sx = (UP ? (x0 >> 1) - 1 : x0 - 1) + t_j[u];
        uint4 val = r_src[u];
        if constexpr (GNIN) {
            ms_f2 v[4];
            ms_unpack8v(val, v);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v[j] = v[j] * ca[j] + cb[j];
                v[j].x = fmaxf(v[j].x, 0.f);
                v[j].y = fmaxf(v[j].y, 0.f);
            }
            val = ms_pack8v(v);                        // the normalised activation is a bf16 tensor in the per-op path
            if (!((unsigned)sy < (unsigned)SH && (unsigned)sx < (unsigned)SW)) val = make_uint4(0, 0, 0, 0);   // outside the image: the convolution's zero padding
        }
        *reinterpret_cast<uint4*>(buf + (size_t)(t_i[u] * SP + t_j[u]) * PX + cc * 8) = val;
    };

#pragma unroll
    for (int u = 0; u < U; ++u) issue(0, u);
    for (int tx = 0; tx < tiles_x; ++tx) {
        const int x0 = tx * TS;
        const int bsel = (tx & 1) * (SP * SP * PX);
        const bf16_t* base_a = base_a0 + bsel;
        const bf16_t* base_b = base_b0 + bsel;
#pragma unroll
        for (int u = 0; u < U; ++u) commit(x0, u, tile[0] + bsel);
        __syncthreads();                                  // tile k is complete; every wave has finished tile k-1, whose buffer tile k+1 will overwrite
        if (tx + 1 < tiles_x) {
#pragma unroll
            for (int u = 0; u < U; ++u) issue(x0 + TS, u);
        }
        // ---- 3x3 convolution: a wave owns 4 rows of the tile, 16 pixels (one row) per MFMA column block ----
        auto conv_row = [&](const int rr) __attribute__((always_inline)) {
            f32x4_t acc[NB];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[nb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            const int y = y0 + wave * 4 + rr, x = x0 + c16;
            const bool live = y < H && x < W;
            // the image's share of this row (lay(adapter(fpn)), L2-resident) is requested BEFORE the taps: its latency used to sit between the last MFMA
            // and the store of every row (waves waited 55-65 % of their life)
            uint2 fr[NB];
            if constexpr (UP) {
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    fr[nb] = make_uint2(0, 0);
                    if (live && nb * 16 + g * 4 < COUT) fr[nb] = *reinterpret_cast<const uint2*>(res_b + ((size_t)y * W + x) * COUT + nb * 16 + g * 4);
                }
            }
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const int dy = t / 3, dx = t % 3;
                    const bf16_t* p = UP ? (dx == 1 ? base_b : base_a) + (((rr + dy + 1) >> 1) * SP + (dx == 0 ? 0 : 1)) * PX
                                         : base_a + ((rr + dy) * HS + dx) * PX;
                    const frag_t a = *reinterpret_cast<const frag_t*>(p + ks * KM);
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) acc[nb] = ms_mfma<KM>(bw[t][nb][ks], a, acc[nb]);
                }
            if constexpr (OUT1) {
                if (live && g == 0) out1[((size_t)n * H + y) * W + x] = acc[0][0] + bs[0][0];
            } else if (live) {
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const int c0 = nb * 16 + g * 4;
                    if (c0 < COUT) {
                        float r0 = acc[nb][0] + bs[nb][0], r1 = acc[nb][1] + bs[nb][1], r2 = acc[nb][2] + bs[nb][2], r3 = acc[nb][3] + bs[nb][3];
                        if constexpr (UP) {           // the image's share: lay(adapter(fpn)) at this pixel
                            r0 += __uint_as_float(fr[nb].x << 16); r1 += __uint_as_float(fr[nb].x & 0xffff0000u);
                            r2 += __uint_as_float(fr[nb].y << 16); r3 += __uint_as_float(fr[nb].y & 0xffff0000u);
                        }
                        const unsigned lo = pack2bf(r0, r1), hi = pack2bf(r2, r3);
                        *reinterpret_cast<uint2*>(out + (((size_t)n * H + y) * W + x) * COUT + c0) = make_uint2(lo, hi);
                        const float v0 = __uint_as_float(lo << 16), v1 = __uint_as_float(lo & 0xffff0000u);
                        const float v2 = __uint_as_float(hi << 16), v3 = __uint_as_float(hi & 0xffff0000u);
                        if constexpr (CG == 2) {
                            ssum[nb][0] += v0 + v1; ssq[nb][0] += v0 * v0 + v1 * v1;
                            ssum[nb][1] += v2 + v3; ssq[nb][1] += v2 * v2 + v3 * v3;
                        } else {
                            ssum[nb][0] += (v0 + v1) + (v2 + v3); ssq[nb][0] += (v0 * v0 + v1 * v1) + (v2 * v2 + v3 * v3);
                        }
                    }
                }
            }
        };
        if constexpr (CIN >= 64) {      // 36 weight fragments live: the four rows share one copy of the tap loop
#pragma unroll 1
            for (int rr = 0; rr < 4; ++rr) conv_row(rr);
        } else {
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) conv_row(rr);
        }
    }
    if constexpr (!OUT1) {
        // ---- statistics of this strip: fold the 16 pixel lanes of a channel quad, then the waves through LDS, then one atomic per (group, moment) ----
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int h = 0; h < (CG == 2 ? 2 : 1); ++h) {
                float s = ssum[nb][h], q = ssq[nb][h];
#pragma unroll
                for (int m = 1; m < 16; m <<= 1) { s += __shfl_xor(s, m); q += __shfl_xor(q, m); }
                const int c0 = nb * 16 + g * 4 + 2 * h;
                if (c16 == 0 && c0 < COUT) {
                    atomicAdd(&sacc[0][c0 / CG], s);
                    atomicAdd(&sacc[1][c0 / CG], q);
                }
            }
        __syncthreads();
        if (tid < 16) atomicAdd(out_stats + ((size_t)n * 8 + (tid & 7)) * 2 + (tid >> 3), sacc[tid >> 3][tid & 7]);
    }
}

template <int CIN, int COUT, bool GNIN, bool UP, bool OUT1>
static void launch_stage(const void* src, const float* src_stats, const float* gamma, const float* beta, const void* fpn_conv, const void* w, const float* bias,
                         void* out, float* out_stats, int N, int Q, int H, int W, int w_rows, float eps, hipStream_t st) {
    const int tiles_y = (H + 15) / 16;
    hipLaunchKernelGGL((mask_stage_kernel<CIN, COUT, GNIN, UP, OUT1>), dim3((unsigned)(N * tiles_y)), dim3(256), 0, st, (const bf16_t*)src, src_stats, gamma, beta,
                       (const bf16_t*)fpn_conv, (const bf16_t*)w, bias, OUT1 ? nullptr : (bf16_t*)out, out_stats, OUT1 ? (float*)out : nullptr, Q, H, W, w_rows, eps);
}

}  // namespace toist

using namespace toist;

// See include/toist_hip.h.  Supported shapes: (c_in, c_out) = (32, 16) with gn_in and up (lay5), (64, 32) with up (lay4), (16, 1) with gn_in, no up (out_lay).
extern "C" int toist_mask_stage_fwd(const void* src, const float* src_stats, const float* gamma, const float* beta, const void* fpn_conv, const void* w,
                                    const float* bias, void* out, float* out_stats, int N, int Q, int H, int W, int c_in, int c_out, int w_rows, int gn_in,
                                    int up, float eps, void* stream) {
    TOIST_REQUIRE(src && w && out && N > 0 && Q > 0 && H > 0 && W > 0 && w_rows > 0, "toist_mask_stage_fwd: bad args");
    TOIST_REQUIRE(!gn_in || (src_stats && gamma && beta), "toist_mask_stage_fwd: gn_in needs src_stats, gamma and beta");
    TOIST_REQUIRE(!up || (fpn_conv && (H % 2) == 0 && (W % 2) == 0), "toist_mask_stage_fwd: up needs fpn_conv and even output sizes");
    TOIST_REQUIRE(c_out == 1 || out_stats, "toist_mask_stage_fwd: out_stats missing");
    TOIST_REQUIRE((long long)N * ((H + 15) / 16) < (1ll << 31), "toist_mask_stage_fwd: too many strips");
    hipStream_t st = (hipStream_t)stream;
    if (c_out != 1) {
        hipError_t e = hipMemsetAsync(out_stats, 0, sizeof(float) * 16 * (size_t)N, st);
        if (e != hipSuccess) { set_last_error("toist_mask_stage_fwd: memset: %s", hipGetErrorString(e)); return TOIST_EHIP; }
    }
    if (c_in == 32 && c_out == 16 && gn_in && up && w_rows == 16)
        launch_stage<32, 16, true, true, false>(src, src_stats, gamma, beta, fpn_conv, w, bias, out, out_stats, N, Q, H, W, w_rows, eps, st);
    else if (c_in == 64 && c_out == 32 && !gn_in && up && w_rows == 32)
        launch_stage<64, 32, false, true, false>(src, src_stats, gamma, beta, fpn_conv, w, bias, out, out_stats, N, Q, H, W, w_rows, eps, st);
    else if (c_in == 16 && c_out == 1 && gn_in && !up && w_rows <= 16)
        launch_stage<16, 16, true, false, true>(src, src_stats, gamma, beta, fpn_conv, w, bias, out, out_stats, N, Q, H, W, w_rows, eps, st);
    else {
        set_last_error("toist_mask_stage_fwd: unsupported stage (c_in %d, c_out %d, w_rows %d, gn_in %d, up %d)", c_in, c_out, w_rows, gn_in, up);
        return TOIST_EINVAL;
    }
    return check_launch("toist_mask_stage_fwd");
}
