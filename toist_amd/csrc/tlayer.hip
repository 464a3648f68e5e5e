// Row-complete sub-layer kernels of the cross-modal transformer (d_model = 256): one launch computes a [16 rows x 256 columns] block of
// an nn.Linear (or of its data gradient) over the WHOLE reduction length and finishes the rows on chip -- bias, dropout, residual and
// LayerNorm in the forward direction; residual gradients, LayerNorm backward and the dropout mask of the branch gradient in the backward
// direction.  Replaces, per transformer layer of /root/reference/models/transformer.py:290-304 (encoder) and :362-408 (decoder),
//   out_proj GEMM + LayerNorm launch, linear2 GEMM (+ split-K fold) + LayerNorm launch                                  (forward)
//   LayerNorm-backward launch + the data-gradient GEMM that produced its input gradient (+ the attention partial fold)   (backward)
//
// Why a 16 x 256 block per workgroup: LayerNorm needs complete rows, so a workgroup must own all 256 output columns of its rows and
// therefore stream the whole weight matrix (128 KB at K = 256, 384 KB at K = 768, 1 MB at K = 2048) through its CU.  The L2 -> CU path
// delivers 25-45 B/clk/CU whatever the load instruction (profiles/r01_load_path_bandwidth_probe.txt), so the launch time is
// |W| / (that rate) however many rows the block has -- few rows per block = many blocks = every CU pulls in parallel (3328 rows ->
// 208 workgroups, 800 rows -> 50).  The kernel is therefore organised around the weight stream:
//   * a wave owns 64 output columns and streams ITS slice of the weights privately -- no operand is shared between waves, so the
//     main loop has no workgroup barrier at all;
//   * weights stored [n][k] (forward) are loaded straight into MFMA B fragments (16 bytes per lane, 4 k-steps = 16 KB per wave in
//     flight); weights stored [k][n] (data gradients read the parameter in place) go through a wave-private LDS tile and come back
//     transposed by ds_read_b64_tr_b16 (no bf16 transposed copies of the parameters exist);
//   * the 16 x K activation block is staged once in LDS (optionally folding the key-split partial sums the attention backward kernel
//     leaves behind, and writing the folded rows back for the weight-gradient GEMM);
//   * the epilogue turns the accumulators through LDS into row pieces of 8 columns: a row lives in one 16-lane group, so the LayerNorm
//     reductions are four shuffles, and every global access is 16 bytes.
#include "common.h"

#include <type_traits>

namespace toist {

namespace {

constexpr int RN = 256;        // output columns = d_model
constexpr int RBM = 16;        // rows per workgroup
constexpr int RCS = RN + 8;    // row stride of the f32 accumulator tile in LDS
constexpr int RTS = 72;        // row stride (bf16) of a wave's [32 k][64 n] weight tile
typedef __attribute__((ext_vector_type(8))) short s16x8_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;

template <int N, typename F>
__device__ __forceinline__ void unrolled(F&& f) {      // f(integral_constant<int, 0>) .. f(integral_constant<int, N - 1>): indices stay compile-time
    if constexpr (N > 0) {
        unrolled<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

__device__ __forceinline__ void unpack8f(const uint4 u, float* v) {
    const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        v[2 * q] = __uint_as_float(w[q] << 16);
        v[2 * q + 1] = __uint_as_float(w[q] & 0xffff0000u);
    }
}
__device__ __forceinline__ uint4 pack8f(const float* v) {
    return make_uint4(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7]));
}
// sum over the 16 lanes of a row group (lanes 16 r .. 16 r + 15 of the wave)
__device__ __forceinline__ float group16_sum(float v) {
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    v += __shfl_xor(v, 8, 64);
    return v;
}

}  // namespace

template <int BKIND, int EPI>
__global__ __launch_bounds__(256) void rowgemm_kernel(const toist_rowgemm_desc p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    typedef __attribute__((address_space(3))) s16x4_t* lds_v4;
    const int K = p.K, AS = K + 8;                        // A rows padded by 16 bytes: 16 rows hit 16 different bank groups
    bf16_t* const sA = reinterpret_cast<bf16_t*>(smem_raw);                         // [16][K + 8]
    bf16_t* const sRing = sA + RBM * AS;                                           // wave-private weight tiles: B_KROW [4][2][32 k][RTS], B_ROWK [4][2][64 n][RTS]
    float* const sC = reinterpret_cast<float*>(smem_raw);                          // after the main loop: [16][RCS] f32
    float* const sRed = sC + RBM * RCS;                                            // LN_BWD: [2][16][256] f32
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c16 = lane & 15;
    const int row0 = blockIdx.x * RBM;
    const int nks = K >> 5;                                // k-steps of 32 (a multiple of 4)

    f32x4_t acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // ---- first weight pieces are requested before anything else: the stream is what the launch waits for ----
    // Both layouts are loaded as WHOLE 128-byte lines (8 lanes per row, 8 rows per wave instruction) and re-laid out through a
    // wave-private LDS tile.  Loading forward weights straight into B fragments (16 rows x 64 bytes per instruction) was measured at
    // half the rate: the second half of every line was fetched from L2 again one k-step later (profiles/r04_rowgemm_us.txt).
    u32x4_t rq[4][4];       // B_KROW: 4 KB chunk (32 k x 64 n) of k-steps s .. s + 3, 8 rows per load
    u32x4_t fq[2][8];       // B_ROWK: 8 KB chunk (64 n x 64 k) of k-step pairs s, s + 1
    const bf16_t* const wl = reinterpret_cast<const bf16_t*>(p.w) + (size_t)(wave * 64 + (lane >> 3)) * p.ldw + (lane & 7) * 8;   // + 8 i rows, + 64 per pair
    const bf16_t* const wk = reinterpret_cast<const bf16_t*>(p.w) + (size_t)(lane >> 3) * p.ldw + wave * 64 + (lane & 7) * 8;
    if (BKIND == TOIST_B_ROWK) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int i = 0; i < 8; ++i) fq[s][i] = *reinterpret_cast<const u32x4_t*>(wl + (size_t)8 * i * p.ldw + s * 64);
    } else {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < 4; ++i) rq[s][i] = *reinterpret_cast<const u32x4_t*>(wk + (size_t)(s * 32 + 8 * i) * p.ldw);
    }

    // ---- activation rows -> LDS (16 threads per row, 16-byte pieces, up to 8 requests in flight per thread) ----
    {
        const int r = tid >> 4, pl = tid & 15, m = row0 + r;
        const bool live = m < p.M;
        const int npc = K >> 3;
        const bool folding = p.fold_parts > 1;
        for (int base = 0; base < npc; base += 128) {
            uint4 v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int pc = base + pl + 16 * i;
                v[i] = make_uint4(0, 0, 0, 0);
                if (live && pc < npc && !(folding && pc * 8 < p.fold_cols))
                    v[i] = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(p.a) + (size_t)m * p.lda + pc * 8);
            }
            if (folding) {
                // columns < fold_cols: the sum of fold_parts bf16 slabs (the key-split partial dQ of the attention backward kernel),
                // added in f32, rounded once, written back to A for the weight-gradient GEMM that reads the same rows later
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int pc = base + pl + 16 * i;
                    if (live && pc < npc && pc * 8 < p.fold_cols) {
                        float s8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                        for (int sp = 0; sp < p.fold_parts; ++sp) {
                            float t8[8];
                            unpack8f(*reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(p.fold) + (size_t)sp * p.fold_stride +
                                                                     (size_t)m * p.fold_cols + pc * 8), t8);
#pragma unroll
                            for (int q = 0; q < 8; ++q) s8[q] += t8[q];
                        }
                        v[i] = pack8f(s8);
                        *reinterpret_cast<uint4*>(const_cast<bf16_t*>(reinterpret_cast<const bf16_t*>(p.a)) + (size_t)m * p.lda + pc * 8) = v[i];
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int pc = base + pl + 16 * i;
                if (pc < npc) *reinterpret_cast<uint4*>(sA + r * AS + pc * 8) = v[i];
            }
        }
    }
    __syncthreads();

    // ---- main loop: no workgroup barrier, every wave streams its own 64 columns of the weights ----
    const bf16_t* const aFrag = sA + c16 * AS + g * 8;        // A fragment of k-step s: + 32 s
    if (BKIND == TOIST_B_ROWK) {
        bf16_t* const tile = sRing + wave * (2 * 64 * RTS);                  // [2][64 n][RTS] (64 k + 8 pad)
        const int wr = (lane >> 3) * RTS + (lane & 7) * 8;                   // where this lane's 16 bytes of an 8-row load land
        const int rd = c16 * RTS + g * 8;                                    // B fragment of column block j, k-step s: + 16 j rows, + 32 s
        for (int kp = 0; kp < nks; kp += 4) {                                // two pairs of k-steps per trip (compile-time buffer indices)
            unrolled<2>([&](auto uu) {
                constexpr int u = decltype(uu)::value;
                bf16_t* const buf = tile + u * (64 * RTS);
                unrolled<8>([&](auto ii) { constexpr int i = decltype(ii)::value; *reinterpret_cast<u32x4_t*>(buf + 8 * i * RTS + wr) = fq[u][i]; });
                if (kp + 2 * u + 4 < nks)
                    unrolled<8>([&](auto ii) {
                        constexpr int i = decltype(ii)::value;
                        fq[u][i] = *reinterpret_cast<const u32x4_t*>(wl + (size_t)8 * i * p.ldw + (size_t)(kp / 2 + u + 2) * 64);
                    });
                unrolled<2>([&](auto ss) {
                    constexpr int s2 = decltype(ss)::value;
                    const bf16x8_t af = *reinterpret_cast<const bf16x8_t*>(aFrag + (kp + 2 * u + s2) * 32);
                    unrolled<4>([&](auto jj) {
                        constexpr int j = decltype(jj)::value;
                        const bf16x8_t bf = *reinterpret_cast<const bf16x8_t*>(buf + rd + 16 * j * RTS + 32 * s2);
                        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bf, acc[j], 0, 0, 0);
                    });
                });
            });
        }
    } else {
        bf16_t* const ring = sRing + wave * (2 * 32 * RTS);
        const int wr = (lane >> 3) * RTS + (lane & 7) * 8;                   // where this lane's 16 bytes of an 8-row load land
        const int rd = (8 * g + (c16 >> 2)) * RTS + (c16 & 3) * 4;           // transposing read: rows 8g .. 8g + 3 (second read: + 4 rows)
        for (int ks = 0; ks < nks; ks += 4) {
            unrolled<4>([&](auto uu) {
                constexpr int u = decltype(uu)::value;
                bf16_t* const buf = ring + (u & 1) * (32 * RTS);
                unrolled<4>([&](auto ii) { constexpr int i = decltype(ii)::value; *reinterpret_cast<u32x4_t*>(buf + 8 * i * RTS + wr) = rq[u][i]; });
                if (ks + u + 4 < nks)
                    unrolled<4>([&](auto ii) {
                        constexpr int i = decltype(ii)::value;
                        rq[u][i] = *reinterpret_cast<const u32x4_t*>(wk + (size_t)((ks + u + 4) * 32 + 8 * i) * p.ldw);
                    });
                const bf16x8_t af = *reinterpret_cast<const bf16x8_t*>(aFrag + (ks + u) * 32);
                unrolled<4>([&](auto jj) {
                    constexpr int j = decltype(jj)::value;
                    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(buf + rd + 16 * j));
                    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(buf + rd + 4 * RTS + 16 * j));
                    const s16x8_t both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, __builtin_bit_cast(bf16x8_t, both), acc[j], 0, 0, 0);
                });
            });
        }
    }
    __syncthreads();        // every wave is done with sA / its ring: the accumulator tile reuses the space

    // ---- accumulators -> LDS [16][256] f32: lane (column c16 of block j, group g) holds rows 4g .. 4g + 3 ----
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) sC[(4 * g + r) * RCS + wave * 64 + j * 16 + c16] = acc[j][r];
    __syncthreads();

    // ---- row layout: thread = (row tid / 16, column pieces tid % 16 and tid % 16 + 16), 8 columns per piece ----
    const int r = tid >> 4, pl = tid & 15, m = row0 + r;
    const bool live = m < p.M;
    const size_t mrow = (size_t)(live ? m : 0);
    float v[2][8];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int c0 = (pl + 16 * h) * 8;
        const float4 lo = *reinterpret_cast<const float4*>(sC + r * RCS + c0), hi = *reinterpret_cast<const float4*>(sC + r * RCS + c0 + 4);
        v[h][0] = lo.x; v[h][1] = lo.y; v[h][2] = lo.z; v[h][3] = lo.w; v[h][4] = hi.x; v[h][5] = hi.y; v[h][6] = hi.z; v[h][7] = hi.w;
        if (p.bias != nullptr) {
            const float4 b0 = *reinterpret_cast<const float4*>(p.bias + c0), b1 = *reinterpret_cast<const float4*>(p.bias + c0 + 4);
            v[h][0] += b0.x; v[h][1] += b0.y; v[h][2] += b0.z; v[h][3] += b0.w; v[h][4] += b1.x; v[h][5] += b1.y; v[h][6] += b1.z; v[h][7] += b1.w;
        }
    }
    unsigned long long seed = p.drop_seed;
    if (p.drop_p > 0.f && p.drop_seed_dev != nullptr) seed += *p.drop_seed_dev;
    const unsigned thresh = p.drop_p > 0.f ? (unsigned)(p.drop_p * 4294967296.0) : 0u;
    const float dscale = p.drop_p > 0.f ? 1.f / (1.f - p.drop_p) : 1.f;
    if (EPI == TOIST_ROW_LN_FWD && p.drop_p > 0.f) {
        // x + dropout(sublayer(x)): the mask is the hash of (seed, element index m * 256 + n), the convention of the GEMM epilogue
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const unsigned long long idx = (unsigned long long)m * RN + (pl + 16 * h) * 8 + q;
                v[h][q] = dropout_keep(seed, idx, thresh) ? v[h][q] * dscale : 0.f;
            }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int c0 = (pl + 16 * h) * 8;
        if (p.res != nullptr && live) {
            float t8[8];
            unpack8f(*reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(p.res) + mrow * p.ldr + c0), t8);
#pragma unroll
            for (int q = 0; q < 8; ++q) v[h][q] += t8[q];
        }
        if (p.res2 != nullptr && live) {
            float t8[8];
            unpack8f(*reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(p.res2) + mrow * p.ldr2 + c0), t8);
#pragma unroll
            for (int q = 0; q < 8; ++q) v[h][q] += t8[q];
        }
    }

    if (EPI == TOIST_ROW_PLAIN) {
        if (live) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
                *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(p.out) + mrow * p.ldo + (pl + 16 * h) * 8) = pack8f(v[h]);
        }
        return;
    }

    if (EPI == TOIST_ROW_LN_FWD) {
        // z = the bf16 value the backward pass will read; the statistics are those of the ROUNDED row (as the stand-alone LayerNorm
        // kernel computes them from the stored sum): two passes, the row lives in this 16-lane group
        float s = 0.f;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const uint4 zp = pack8f(v[h]);
            if (live && p.z != nullptr) *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(p.z) + mrow * RN + (pl + 16 * h) * 8) = zp;
            unpack8f(zp, v[h]);
#pragma unroll
            for (int q = 0; q < 8; ++q) s += v[h][q];
        }
        const float mean = group16_sum(s) * (1.f / RN);
        float qq = 0.f;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int q = 0; q < 8; ++q) { const float d = v[h][q] - mean; qq += d * d; }
        const float rstd = rsqrtf(group16_sum(qq) * (1.f / RN) + p.eps);
        if (live) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int c0 = (pl + 16 * h) * 8;
                float gm[8], bt[8], o[8];
                *reinterpret_cast<float4*>(gm) = *reinterpret_cast<const float4*>(p.gamma + c0);
                *reinterpret_cast<float4*>(gm + 4) = *reinterpret_cast<const float4*>(p.gamma + c0 + 4);
                *reinterpret_cast<float4*>(bt) = *reinterpret_cast<const float4*>(p.beta + c0);
                *reinterpret_cast<float4*>(bt + 4) = *reinterpret_cast<const float4*>(p.beta + c0 + 4);
#pragma unroll
                for (int q = 0; q < 8; ++q) o[q] = (v[h][q] - mean) * rstd * gm[q] + bt[q];
                const uint4 yo = pack8f(o);
                *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(p.out) + mrow * p.ldo + c0) = yo;
                if (p.out2 != nullptr) {       // bf16(bf16(y) + add): what the stand-alone add kernel computed from the stored y
                    float a8[8], y8[8];
                    unpack8f(*reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(p.add) + mrow * RN + c0), a8);
                    unpack8f(yo, y8);
#pragma unroll
                    for (int q = 0; q < 8; ++q) y8[q] += a8[q];
                    *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(p.out2) + mrow * RN + c0) = pack8f(y8);
                }
            }
            if (pl == 0) {
                if (p.mean != nullptr) p.mean[m] = mean;
                if (p.rstd != nullptr) p.rstd[m] = rstd;
            }
        }
        return;
    }

    // ---- LN_BWD: v = gradient w.r.t. the LayerNorm output (data gradient + residual gradients, f32) ----
    {
        const float mu = live ? p.mean[m] : 0.f, rs = live ? p.rstd[m] : 0.f;
        float xh[2][8], gg[2][8];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int c0 = (pl + 16 * h) * 8;
            float z8[8], gm[8];
            unpack8f(live ? *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(p.z) + mrow * RN + c0) : make_uint4(0, 0, 0, 0), z8);
            *reinterpret_cast<float4*>(gm) = *reinterpret_cast<const float4*>(p.gamma + c0);
            *reinterpret_cast<float4*>(gm + 4) = *reinterpret_cast<const float4*>(p.gamma + c0 + 4);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if (!live) v[h][q] = 0.f;
                xh[h][q] = (z8[q] - mu) * rs;
                gg[h][q] = v[h][q] * gm[q];
                s1 += gg[h][q];
                s2 += gg[h][q] * xh[h][q];
            }
        }
        s1 = group16_sum(s1) * (1.f / RN);
        s2 = group16_sum(s2) * (1.f / RN);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int c0 = (pl + 16 * h) * 8;
            float o[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) o[q] = rs * (gg[h][q] - s1 - xh[h][q] * s2);
            if (live) {
                *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(p.out) + mrow * p.ldo + c0) = pack8f(o);
                if (p.out2 != nullptr) {       // the gradient of the dropout(branch) term: same mask as the forward pass
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const unsigned long long idx = (unsigned long long)m * RN + c0 + q;
                        o[q] = dropout_keep(seed, idx, thresh) ? o[q] * dscale : 0.f;
                    }
                    *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(p.out2) + mrow * RN + c0) = pack8f(o);
                }
            }
            if (p.partials != nullptr) {
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    sRed[r * RN + c0 + q] = v[h][q] * xh[h][q];
                    sRed[RBM * RN + r * RN + c0 + q] = v[h][q];
                }
            }
        }
        if (p.partials != nullptr) {
            // sRed lies behind sC: no thread is still reading sC rows it does not own, but the sums below cross rows
            __syncthreads();
            float sg = 0.f, sb = 0.f;
#pragma unroll
            for (int rr = 0; rr < RBM; ++rr) {
                sg += sRed[rr * RN + tid];
                sb += sRed[RBM * RN + rr * RN + tid];
            }
            p.partials[(size_t)blockIdx.x * RN + tid] = sg;                        // [2][blocks][256]: folded by toist_splitk_reduce_batch
            p.partials[((size_t)gridDim.x + blockIdx.x) * RN + tid] = sb;
        }
    }
}

}  // namespace toist

using namespace toist;

extern "C" int toist_rowgemm_blocks(int M) { return M > 0 ? (M + RBM - 1) / RBM : 0; }

extern "C" int toist_rowgemm(const toist_rowgemm_desc* d, void* stream) {
    TOIST_REQUIRE(d != nullptr && d->M > 0 && d->K >= 128 && (d->K % 128) == 0 && d->K <= 4096, "toist_rowgemm: M=%d K=%d (K a multiple of 128, <= 4096)",
                  d ? d->M : 0, d ? d->K : 0);
    TOIST_REQUIRE(d->a && d->w && d->out, "toist_rowgemm: a, w and out are required");
    TOIST_REQUIRE(d->b_kind == TOIST_B_ROWK || d->b_kind == TOIST_B_KROW, "toist_rowgemm: b_kind %d", d->b_kind);
    TOIST_REQUIRE(d->epi >= TOIST_ROW_PLAIN && d->epi <= TOIST_ROW_LN_BWD, "toist_rowgemm: epi %d", d->epi);
    TOIST_REQUIRE((d->lda % 8) == 0 && d->lda >= d->K && (d->ldw % 8) == 0 && d->ldw >= (d->b_kind == TOIST_B_ROWK ? d->K : RN) && (d->ldo % 8) == 0 && d->ldo >= RN,
                  "toist_rowgemm: leading dimensions must be multiples of 8 elements and cover the rows (lda %d, ldw %d, ldo %d)", d->lda, d->ldw, d->ldo);
    TOIST_REQUIRE(((((size_t)d->a) | ((size_t)d->w) | ((size_t)d->out) | ((size_t)d->res) | ((size_t)d->res2) | ((size_t)d->z) | ((size_t)d->add) | ((size_t)d->out2) |
                    ((size_t)d->fold) | ((size_t)d->bias) | ((size_t)d->gamma) | ((size_t)d->beta)) & 15) == 0, "toist_rowgemm: every pointer must be 16-byte aligned");
    TOIST_REQUIRE(d->res == nullptr || ((d->ldr % 8) == 0 && d->ldr >= RN), "toist_rowgemm: ldr");
    TOIST_REQUIRE(d->res2 == nullptr || ((d->ldr2 % 8) == 0 && d->ldr2 >= RN), "toist_rowgemm: ldr2");
    TOIST_REQUIRE(d->fold_parts <= 1 || (d->fold != nullptr && d->fold_cols > 0 && (d->fold_cols % 8) == 0 && d->fold_cols <= d->K && d->fold_parts <= 16),
                  "toist_rowgemm: fold needs slabs, fold_cols %% 8 == 0 and <= K, at most 16 parts");
    TOIST_REQUIRE(d->drop_p >= 0.f && d->drop_p < 1.f, "toist_rowgemm: bad dropout p");
    if (d->epi == TOIST_ROW_LN_FWD) {
        TOIST_REQUIRE(d->gamma && d->beta, "toist_rowgemm: LN_FWD needs gamma and beta");
        TOIST_REQUIRE((d->add == nullptr) == (d->out2 == nullptr), "toist_rowgemm: add and out2 come together");
    }
    if (d->epi == TOIST_ROW_LN_BWD) {
        TOIST_REQUIRE(d->gamma && d->z && d->mean && d->rstd, "toist_rowgemm: LN_BWD needs gamma, z, mean and rstd");
        TOIST_REQUIRE(d->out2 == nullptr || d->drop_p > 0.f, "toist_rowgemm: LN_BWD out2 is the dropout-masked gradient (drop_p > 0)");
    }
    const size_t sz_a = (size_t)RBM * (d->K + 8) * sizeof(bf16_t) + (size_t)4 * 2 * (d->b_kind == TOIST_B_KROW ? 32 : 64) * RTS * sizeof(bf16_t);
    const size_t sz_e = (size_t)RBM * RCS * sizeof(float) + (d->epi == TOIST_ROW_LN_BWD ? (size_t)2 * RBM * RN * sizeof(float) : 0);
    const size_t lds = sz_a > sz_e ? sz_a : sz_e;
    const dim3 grid((d->M + RBM - 1) / RBM), block(256);
    hipStream_t st = (hipStream_t)stream;
#define TOIST_ROWGEMM(BK, EP, SLOT)                                                                                                        \
    do {                                                                                                                                   \
        if (lds > 64 * 1024) {                                                                                                             \
            const bool ok = lds_attr_once(SLOT, [&]() {                                                                                    \
                return hipFuncSetAttribute((const void*)rowgemm_kernel<BK, EP>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess; \
            });                                                                                                                            \
            if (!ok) { set_last_error("toist_rowgemm: cannot raise the dynamic LDS limit"); return TOIST_EHIP; }                          \
        }                                                                                                                                  \
        hipLaunchKernelGGL((rowgemm_kernel<BK, EP>), grid, block, lds, st, *d);                                                            \
    } while (0)
    // lds_attr_once keeps 8 flag words: tlayer uses slots 2 .. 7
    if (d->b_kind == TOIST_B_ROWK) {
        if (d->epi == TOIST_ROW_PLAIN) TOIST_ROWGEMM(TOIST_B_ROWK, TOIST_ROW_PLAIN, 2);
        else if (d->epi == TOIST_ROW_LN_FWD) TOIST_ROWGEMM(TOIST_B_ROWK, TOIST_ROW_LN_FWD, 3);
        else TOIST_ROWGEMM(TOIST_B_ROWK, TOIST_ROW_LN_BWD, 4);
    } else {
        if (d->epi == TOIST_ROW_PLAIN) TOIST_ROWGEMM(TOIST_B_KROW, TOIST_ROW_PLAIN, 5);
        else if (d->epi == TOIST_ROW_LN_FWD) TOIST_ROWGEMM(TOIST_B_KROW, TOIST_ROW_LN_FWD, 6);
        else TOIST_ROWGEMM(TOIST_B_KROW, TOIST_ROW_LN_BWD, 7);
    }
#undef TOIST_ROWGEMM
    return check_launch("toist_rowgemm");
}
