// 3x3 / stride 1 / pad 1 convolutions with very few channels (the 160x160 stages of the mask head: 32->16 and
// 16->1 channels over 20 M pixels, /root/reference/models/segmentation.py:176-241 lay5 / out_lay) -- HBM-bound work.
// A tile-and-stage implicit GEMM spends its time staging 64-wide k-tiles that are mostly padding; here one wavefront
// owns 16 consecutive pixels per iteration and feeds the MFMA straight from global memory:
//   A fragment of a tap = the 16 pixels shifted by the tap, KC channels each: lane (pixel c16, channel group g) loads
//   its 8 (KC = 32) or 4 (KC <= 16) channels with ONE 16- / 8-byte load -- a wave reads 16 px x KC x 2 B contiguous;
//   the nine taps re-read rows that are in L1/L2, so HBM sees the input once;
//   B fragments (the 3x3xKCxN weights) live in registers for the whole kernel;
//   D^T = B.A^T leaves every lane with 4 consecutive output channels of one pixel: 8-byte stores, 16 px contiguous.
// forward : src = x [P,KC],  w[co][tap][ci]            (tap offset dy = r-1, dx = s-1)
// dgrad   : src = dy [P,KC], w[co][tap][ci] transposed  (dy = 1-r, dx = 1-s), KC = output channels of the forward conv
#include "common.h"

namespace toist {

typedef __attribute__((ext_vector_type(4))) short bf16x4s_t;

template <int KC>
struct Frag {  // MFMA operand of one lane
    typedef typename std::conditional<KC == 32, bf16x8_t, bf16x4s_t>::type type;
};

template <int KC>
__device__ __forceinline__ f32x4_t mfma_kc(typename Frag<KC>::type b, typename Frag<KC>::type a, f32x4_t c) {
    if constexpr (KC == 32) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a, c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(b, a, c, 0, 0, 0);
}

// KC: source channels per tap (8, 16 or 32; 8 runs on the 16-wide MFMA with the upper half zero).  NB: 16-column output blocks.
template <int KC, int NB, bool DGRAD>
__global__ __launch_bounds__(256) void conv3_small_kernel(const bf16_t* __restrict__ src, const bf16_t* __restrict__ w, const float* __restrict__ shift,
                                                          const bf16_t* __restrict__ res, bf16_t* __restrict__ out, int NIMG, int H, int W, int NOUT,
                                                          int w_co, int w_ci) {
    constexpr int KL = (KC == 32) ? 8 : 4;             // channels per lane
    constexpr int KM = (KC == 32) ? 32 : 16;           // MFMA reduction width
    typedef typename Frag<KC>::type frag_t;
    const int lane = threadIdx.x & 63, g = lane >> 4, c16 = lane & 15;
    const long long P = (long long)NIMG * H * W;

    // ---- weights -> registers: B[n][k] of tap t, column block nb ----
    frag_t bw[9][NB];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            union { bf16_t raw[KL]; frag_t v; } f;          // raw bf16 bits (an element assignment would convert numerically)
#pragma unroll
            for (int j = 0; j < KL; ++j) {
                const int n = nb * 16 + c16, kk = g * KL + j;
                bf16_t v = 0;
                if (kk < KC && n < NOUT) {
                    // forward: n = co, k = ci -> w[n][t][kk];  dgrad: n = ci, k = co -> w[kk][t][n]
                    v = DGRAD ? w[((long long)kk * 9 + t) * w_ci + n] : w[((long long)n * 9 + t) * w_ci + kk];
                }
                f.raw[j] = v;
            }
            bw[t][nb] = f.v;
        }
    (void)w_co;
    float sh[NB][4];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = nb * 16 + g * 4 + j;
            sh[nb][j] = (shift != nullptr && n < NOUT) ? shift[n] : 0.f;
        }

    const long long groups = (P + 15) >> 4;
    const long long wave_id = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long long)gridDim.x * 4;
    const int HW = H * W;
    for (long long grp = wave_id; grp < groups; grp += nwaves) {
        const long long p = (grp << 4) + c16;
        const bool live = p < P;
        const int rem = (int)((live ? p : 0) % HW);
        const int y = rem / W, x = rem - y * W;
        f32x4_t acc[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[nb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        frag_t a[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int r = t / 3, s_ = t - r * 3;
            const int dy = DGRAD ? 1 - r : r - 1, dx = DGRAD ? 1 - s_ : s_ - 1;
            const bool ok = live && (unsigned)(y + dy) < (unsigned)H && (unsigned)(x + dx) < (unsigned)W && (g * KL < KC);
            union { bf16_t raw[KL]; frag_t v; } f;
#pragma unroll
            for (int j = 0; j < KL; ++j) f.raw[j] = 0;
            if (ok) f.v = *reinterpret_cast<const frag_t*>(src + (p + (long long)dy * W + dx) * KC + g * KL);
            a[t] = f.v;
        }
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[nb] = mfma_kc<KC>(bw[t][nb], a[t], acc[nb]);
        if (live) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const int n = nb * 16 + g * 4;
                if (n < NOUT) {
                    float v0 = acc[nb][0] + sh[nb][0], v1 = acc[nb][1] + sh[nb][1], v2 = acc[nb][2] + sh[nb][2], v3 = acc[nb][3] + sh[nb][3];
                    const long long o = p * NOUT + n;
                    if (res != nullptr) {
                        const uint2 u = *reinterpret_cast<const uint2*>(res + o);
                        v0 += __uint_as_float(u.x << 16); v1 += __uint_as_float(u.x & 0xffff0000u);
                        v2 += __uint_as_float(u.y << 16); v3 += __uint_as_float(u.y & 0xffff0000u);
                    }
                    *reinterpret_cast<uint2*>(out + o) = make_uint2(pack2bf(v0, v1), pack2bf(v2, v3));
                }
            }
        }
    }
}

// Weight gradient of the same convolutions: dW[co][tap][ci] = sum_p dy[p, co] * x[p + tap, ci], a reduction over all
// pixels (K = 20 M) onto a 16 x (9*C) result.  One wavefront owns 32 consecutive pixels per iteration: the dy block
// [32 px][CO] and, tap row by tap row, the three shifted x blocks [32 px][C] go global -> registers -> (wave-private)
// LDS, from where both MFMA operands are read k-major (pixel = reduction index) with ds_read_b64_tr_b16; the 9*C/16
// accumulator tiles stay in registers for the whole kernel.  The four waves of a workgroup fold through LDS and every
// workgroup writes one fp32 partial [CO][9*C] that toist_splitk_reduce_batch sums into the gradient.
template <int C, int CO>
__global__ __launch_bounds__(256) void wgrad3_small_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x, float* __restrict__ ws,
                                                           int NIMG, int H, int W) {
    constexpr int NB = C / 16;                  // 16-wide blocks of input channels
    constexpr int XCH = 32 * C / 8 / 64;        // 16-byte chunks of one x block per lane (2 for C = 32, 1 for C = 16)
    constexpr int DCH = (32 * CO / 8 + 63) / 64;
    constexpr int XBLK = 32 * C, DBLK = 32 * CO;   // elements
    __shared__ __attribute__((aligned(16))) bf16_t smem[4][DBLK + 3 * XBLK];
    __shared__ float fold[4][16 * 16];
    typedef __attribute__((address_space(3))) s16x4_t* lds_v4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, c16 = lane & 15;
    bf16_t* sD = smem[wave];
    bf16_t* sX = sD + DBLK;
    const long long P = (long long)NIMG * H * W;
    const int HW = H * W;

    f32x4_t acc[9][NB];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[t][nb] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // k-major fragment of a [32 px][ROWS] block: lane (row block r0 + c16, pixels 8g .. 8g+7)
    auto frag = [&](const bf16_t* blk, const int ROWS, const int r0) -> bf16x8_t {
        const int kpx = 8 * g + (c16 >> 2);
        int col = r0 + ((c16 & 3) >> 1) * 8 + (c16 & 1) * 4;
        if (col >= ROWS) col -= 8;   // CO = 8: result rows 8..15 do not exist -- every lane still takes part in the transposing
                                     // read (they duplicate rows 0..7; those result rows are never stored)
        union { struct { s16x4_t a, b; } h; bf16x8_t v; } u;
        u.h.a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(blk + kpx * ROWS + col));
        u.h.b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(blk + (kpx + 4) * ROWS + col));
        return u.v;
    };

    const long long blocks = (P + 31) >> 5;
    const long long wave_id = (long long)blockIdx.x * 4 + wave, nwaves = (long long)gridDim.x * 4;
    for (long long b = wave_id; b < blocks; b += nwaves) {
        const long long p0 = b << 5;
        // dy block -> LDS
#pragma unroll
        for (int i = 0; i < DCH; ++i) {
            const int c = lane + 64 * i;
            if (c < 32 * CO / 8) {
                const long long e = p0 * CO + (long long)c * 8;
                uint4 v = make_uint4(0, 0, 0, 0);
                if (p0 + c / (CO / 8) < P) v = *reinterpret_cast<const uint4*>(dy + e);
                *reinterpret_cast<uint4*>(sD + c * 8) = v;
            }
        }
        // image coordinates of the pixels this lane stages
        int py[XCH], px[XCH];
        bool pin[XCH];
#pragma unroll
        for (int i = 0; i < XCH; ++i) {
            const long long q = p0 + (lane + 64 * i) / (C / 8);
            pin[i] = q < P;
            const int rem = (int)((pin[i] ? q : 0) % HW);
            py[i] = rem / W;
            px[i] = rem - py[i] * W;
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            uint4 stage[3][XCH];
#pragma unroll
            for (int s_ = 0; s_ < 3; ++s_)
#pragma unroll
                for (int i = 0; i < XCH; ++i) {
                    const int c = lane + 64 * i;
                    const bool ok = pin[i] && (unsigned)(py[i] + r - 1) < (unsigned)H && (unsigned)(px[i] + s_ - 1) < (unsigned)W;
                    uint4 v = make_uint4(0, 0, 0, 0);
                    if (ok) v = *reinterpret_cast<const uint4*>(x + (p0 + (long long)(r - 1) * W + (s_ - 1)) * C + (long long)c * 8);
                    stage[s_][i] = v;
                }
            __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the previous tap row's fragment reads are done before the block is overwritten
#pragma unroll
            for (int s_ = 0; s_ < 3; ++s_)
#pragma unroll
                for (int i = 0; i < XCH; ++i) *reinterpret_cast<uint4*>(sX + s_ * XBLK + (lane + 64 * i) * 8) = stage[s_][i];
            __builtin_amdgcn_s_waitcnt(0xc07f);
            const bf16x8_t a = frag(sD, CO, 0);
#pragma unroll
            for (int s_ = 0; s_ < 3; ++s_)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
                    acc[r * 3 + s_][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag(sX + s_ * XBLK, C, nb * 16), a, acc[r * 3 + s_][nb], 0, 0, 0);
        }
    }
    // fold the four waves, one 16x16 tile at a time, and write this workgroup's partial: ws[blockIdx][co][tap*C + ci]
    float* out = ws + (size_t)blockIdx.x * CO * 9 * C;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            __syncthreads();
            *reinterpret_cast<f32x4_t*>(&fold[wave][c16 * 16 + g * 4]) = acc[t][nb];
            __syncthreads();
            const int e = threadIdx.x;   // 256 threads = the 16 x 16 tile
            const int co = e >> 4, ci = e & 15;
            if (co < CO) out[(size_t)co * 9 * C + t * C + nb * 16 + ci] = (fold[0][e] + fold[1][e]) + (fold[2][e] + fold[3][e]);
        }
}

template <int KC, int NB>
static void launch_small(bool dgrad, const void* src, const void* w, const float* shift, const void* res, void* out, int NIMG, int H, int W, int NOUT,
                         int w_co, int w_ci, hipStream_t st) {
    const long long groups = ((long long)NIMG * H * W + 15) / 16;
    long long blocks = (groups + 3) / 4;
    if (blocks > 256 * 8) blocks = 256 * 8;            // persistent: 8 workgroups per CU, waves stride over the pixel groups
    if (dgrad)
        hipLaunchKernelGGL((conv3_small_kernel<KC, NB, true>), dim3((unsigned)blocks), dim3(256), 0, st, (const bf16_t*)src, (const bf16_t*)w, shift,
                           (const bf16_t*)res, (bf16_t*)out, NIMG, H, W, NOUT, w_co, w_ci);
    else
        hipLaunchKernelGGL((conv3_small_kernel<KC, NB, false>), dim3((unsigned)blocks), dim3(256), 0, st, (const bf16_t*)src, (const bf16_t*)w, shift,
                           (const bf16_t*)res, (bf16_t*)out, NIMG, H, W, NOUT, w_co, w_ci);
}

}  // namespace toist

using namespace toist;

extern "C" int toist_conv3x3_small(int dgrad, const void* src, const void* w, const float* shift, const void* res, void* out, int n_img, int H, int W,
                                   int c_src, int c_out, int w_co, int w_ci, void* stream) {
    TOIST_REQUIRE(src && w && out && n_img > 0 && H > 0 && W > 0, "toist_conv3x3_small: bad args");
    TOIST_REQUIRE(c_src == 8 || c_src == 16 || c_src == 32, "toist_conv3x3_small: source channels must be 8, 16 or 32 (got %d)", c_src);
    TOIST_REQUIRE(c_out > 0 && c_out <= 32 && (c_out % 4) == 0, "toist_conv3x3_small: output channels must be a multiple of 4, <= 32 (got %d)", c_out);
    TOIST_REQUIRE(dgrad ? (w_co == c_src && w_ci == c_out) : (w_ci == c_src && w_co == c_out), "toist_conv3x3_small: weight shape does not match");
    hipStream_t st = (hipStream_t)stream;
    const bool two = c_out > 16;
#define TOIST_SMALL(KC)                                                                                                     \
    do {                                                                                                                    \
        if (two) launch_small<KC, 2>(dgrad != 0, src, w, shift, res, out, n_img, H, W, c_out, w_co, w_ci, st);             \
        else launch_small<KC, 1>(dgrad != 0, src, w, shift, res, out, n_img, H, W, c_out, w_co, w_ci, st);                 \
    } while (0)
    if (c_src == 32) TOIST_SMALL(32);
    else if (c_src == 16) TOIST_SMALL(16);
    else TOIST_SMALL(8);
#undef TOIST_SMALL
    return check_launch("toist_conv3x3_small");
}

extern "C" int toist_wgrad3x3_small_blocks(void) { return 256 * 2; }

// ws must hold toist_wgrad3x3_small_blocks() * c_out * 9 * c_in floats; the caller folds the partials with
// toist_splitk_reduce_batch (splits = blocks, M = c_out, N = 9 * c_in).
extern "C" int toist_wgrad3x3_small(const void* dy, const void* x, float* ws, int n_img, int H, int W, int c_in, int c_out, void* stream) {
    TOIST_REQUIRE(dy && x && ws && n_img > 0 && H > 0 && W > 0, "toist_wgrad3x3_small: bad args");
    TOIST_REQUIRE((c_in == 16 || c_in == 32) && (c_out == 8 || c_out == 16), "toist_wgrad3x3_small: supports 16/32 input and 8/16 output channels (got %d, %d)",
                  c_in, c_out);
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(toist_wgrad3x3_small_blocks()), block(256);
    if (c_in == 32 && c_out == 16) hipLaunchKernelGGL((wgrad3_small_kernel<32, 16>), grid, block, 0, st, (const bf16_t*)dy, (const bf16_t*)x, ws, n_img, H, W);
    else if (c_in == 32) hipLaunchKernelGGL((wgrad3_small_kernel<32, 8>), grid, block, 0, st, (const bf16_t*)dy, (const bf16_t*)x, ws, n_img, H, W);
    else if (c_out == 16) hipLaunchKernelGGL((wgrad3_small_kernel<16, 16>), grid, block, 0, st, (const bf16_t*)dy, (const bf16_t*)x, ws, n_img, H, W);
    else hipLaunchKernelGGL((wgrad3_small_kernel<16, 8>), grid, block, 0, st, (const bf16_t*)dy, (const bf16_t*)x, ws, n_img, H, W);
    return check_launch("toist_wgrad3x3_small");
}
