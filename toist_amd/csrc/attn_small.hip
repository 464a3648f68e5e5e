// Whole-head attention for SHORT sequences (the text encoder: RoBERTa self-attention over a 16-token caption, 12 heads of 64
// features; /root/reference/models/transformer.py:129-130 through transformers' RobertaSelfAttention): S <= 64 keys, head dim <= 64.
//
// At S = 16 a head is 16 x 16 scores: the tiled path spent three launches forward (batched score GEMM, masked softmax, context GEMM)
// and five backward (four batched GEMMs + softmax backward), every one a launch-latency floor with 96 sixteen-row problems.  Here
// one workgroup owns one (caption, head): q, k, v go to LDS as fp32, scores / softmax / dropout / context are plain fp32 FMA loops
// over LDS (82 K multiply-adds per head at S = 16 -- MFMA tiles would be 94 % padding), and the backward kernel re-forms the
// probabilities from the stored row statistics (max, 1 / sum) and the dropout mask from the (seed, element) hash, so nothing
// score-shaped ever reaches HBM.  Latency-bound by construction (96 workgroups): neither roofline applies.
#include "common.h"

namespace toist {

constexpr int AS_MAXS = 64, AS_MAXD = 64, AS_THREADS = 256;

constexpr int AS_LD = AS_MAXD + 4;       // fp32 row pitch: 16-byte aligned rows (float4 reads), consecutive rows 4 banks apart
struct AsShared {
    float q[AS_MAXS][AS_LD];
    float k[AS_MAXS][AS_LD];
    float v[AS_MAXS][AS_LD];
    float p[AS_MAXS][AS_MAXS + 1];      // scores -> probabilities (-> dS in backward)
    float pd[AS_MAXS][AS_MAXS + 1];     // dropped-out probabilities (backward: also dP)
    float g[AS_MAXS][AS_LD];            // backward: dO
    float rowdot[AS_MAXS];
    unsigned char dead[AS_MAXS];
};

// rows of one head -> LDS as fp32; `bias` (optional, fp32 [H * dh]) is the projection bias, added here so that the packed q | k | v
// projection can run as one plain GEMM without an epilogue vector (the three biases are separate parameters)
__device__ __forceinline__ void as_load(float (*dst)[AS_LD], const bf16_t* src, int ld, int b, int h, int S, int dh, const float* bias) {
    const int per = dh >> 3;                                    // 16-byte chunks per row
    for (int c = threadIdx.x; c < S * per; c += AS_THREADS) {
        const int i = c / per, ch = c - i * per;
        const uint4 u = *reinterpret_cast<const uint4*>(src + ((size_t)b * S + i) * ld + h * dh + ch * 8);
        const unsigned w[4] = {u.x, u.y, u.z, u.w};
        float4 blo = make_float4(0.f, 0.f, 0.f, 0.f), bhi = blo;
        if (bias) {
            blo = *reinterpret_cast<const float4*>(bias + h * dh + ch * 8);
            bhi = *reinterpret_cast<const float4*>(bias + h * dh + ch * 8 + 4);
        }
        *reinterpret_cast<float4*>(&dst[i][ch * 8]) = make_float4(__uint_as_float(w[0] << 16) + blo.x, __uint_as_float(w[0] & 0xffff0000u) + blo.y,
                                                                  __uint_as_float(w[1] << 16) + blo.z, __uint_as_float(w[1] & 0xffff0000u) + blo.w);
        *reinterpret_cast<float4*>(&dst[i][ch * 8 + 4]) = make_float4(__uint_as_float(w[2] << 16) + bhi.x, __uint_as_float(w[2] & 0xffff0000u) + bhi.y,
                                                                      __uint_as_float(w[3] << 16) + bhi.z, __uint_as_float(w[3] & 0xffff0000u) + bhi.w);
    }
}

// sum_c a[c] * b[c] over dh (a multiple of 8) features of two LDS rows, four at a time
__device__ __forceinline__ float as_dot(const float* a, const float* b, int dh) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
    for (int c = 0; c < dh; c += 4) {
        const float4 x = *reinterpret_cast<const float4*>(a + c), y = *reinterpret_cast<const float4*>(b + c);
        acc.x += x.x * y.x; acc.y += x.y * y.y; acc.z += x.z * y.z; acc.w += x.w * y.w;
    }
    return (acc.x + acc.y) + (acc.z + acc.w);
}

// scores, masked softmax and dropout of one head into sh.p (probabilities) and sh.pd (dropped-out, rescaled); FWD also stores
// the row statistics, BWD reads them so that p is re-formed exactly as the forward kernel formed it
template <bool FWD>
__device__ __forceinline__ void as_probabilities(AsShared& sh, const unsigned char* key_pad, int b, int bh, int S, int dh, float scale, float drop_p,
                                                 unsigned long long seed, float* stats) {
    for (int j = threadIdx.x; j < S; j += AS_THREADS) sh.dead[j] = (key_pad != nullptr && key_pad[(size_t)b * S + j]) ? 1 : 0;
    __syncthreads();
    for (int e = threadIdx.x; e < S * S; e += AS_THREADS) {
        const int i = e / S, j = e - i * S;
        const float acc = as_dot(sh.q[i], sh.k[j], dh);
        sh.p[i][j] = sh.dead[j] ? -INFINITY : acc * scale;
    }
    __syncthreads();
    const unsigned thresh = drop_p > 0.f ? (unsigned)(drop_p * 4294967296.0) : 0u;
    const float dscale = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    // a row of scores is spread over G = 16 / 32 / 64 lanes of one wavefront (one element per lane; row statistics by xor-shuffles)
    const int G = S <= 16 ? 16 : (S <= 32 ? 32 : 64);
    const int jl = threadIdx.x & (G - 1);
    for (int i = threadIdx.x / G; i < S; i += AS_THREADS / G) {
        const float sc = jl < S ? sh.p[i][jl] : -INFINITY;
        float mx, rs;
        if (FWD) {
            mx = sc;
            for (int o = G >> 1; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
            float sum = jl < S ? __expf(sc - mx) : 0.f;
            for (int o = G >> 1; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
            rs = 1.f / sum;
            if (jl == 0) {
                stats[2 * ((size_t)bh * S + i)] = mx;
                stats[2 * ((size_t)bh * S + i) + 1] = rs;
            }
        } else {
            mx = stats[2 * ((size_t)bh * S + i)];
            rs = stats[2 * ((size_t)bh * S + i) + 1];
        }
        if (jl < S) {
            const float pv = __expf(sc - mx) * rs;
            const bool keep = drop_p <= 0.f || dropout_keep(seed, ((unsigned long long)bh * S + i) * S + jl, thresh);
            sh.p[i][jl] = pv;
            sh.pd[i][jl] = keep ? pv * dscale : 0.f;
        }
    }
    __syncthreads();
}

__global__ __launch_bounds__(AS_THREADS) void attn_small_fwd_kernel(const bf16_t* __restrict__ q, int ldq, const bf16_t* __restrict__ kmat, int ldk,
                                                                    const bf16_t* __restrict__ v, int ldv, const unsigned char* __restrict__ key_pad,
                                                                    int H, int S, int dh, float scale, float drop_p, unsigned long long seed,
                                                                    const unsigned long long* __restrict__ seed_dev, bf16_t* __restrict__ ctx, int ldo,
                                                                    float* __restrict__ stats, const float* __restrict__ bq, const float* __restrict__ bk,
                                                                    const float* __restrict__ bv) {
    extern __shared__ __attribute__((aligned(16))) unsigned char as_raw[];
    AsShared& sh = *reinterpret_cast<AsShared*>(as_raw);
    if (seed_dev) seed += *seed_dev;
    const int bh = blockIdx.x, b = bh / H, h = bh - b * H;
    as_load(sh.q, q, ldq, b, h, S, dh, bq);
    as_load(sh.k, kmat, ldk, b, h, S, dh, bk);
    as_load(sh.v, v, ldv, b, h, S, dh, bv);
    __syncthreads();
    as_probabilities<true>(sh, key_pad, b, bh, S, dh, scale, drop_p, seed, stats);
    const int quarter = dh >> 2;
    for (int e = threadIdx.x; e < S * quarter; e += AS_THREADS) {      // four features of one query row per thread
        const int i = e / quarter, c = (e - i * quarter) * 4;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int j = 0; j < S; ++j) {
            const float w = sh.pd[i][j];
            const float4 x = *reinterpret_cast<const float4*>(&sh.v[j][c]);
            a.x += w * x.x; a.y += w * x.y; a.z += w * x.z; a.w += w * x.w;
        }
        *reinterpret_cast<uint2*>(ctx + ((size_t)b * S + i) * ldo + h * dh + c) = make_uint2(pack2bf(a.x, a.y), pack2bf(a.z, a.w));
    }
}

__global__ __launch_bounds__(AS_THREADS) void attn_small_bwd_kernel(const bf16_t* __restrict__ q, int ldq, const bf16_t* __restrict__ kmat, int ldk,
                                                                    const bf16_t* __restrict__ v, int ldv, const unsigned char* __restrict__ key_pad,
                                                                    int H, int S, int dh, float scale, float drop_p, unsigned long long seed,
                                                                    const unsigned long long* __restrict__ seed_dev, const float* __restrict__ stats,
                                                                    const bf16_t* __restrict__ dctx, int lddo, bf16_t* __restrict__ dq, int lddq,
                                                                    bf16_t* __restrict__ dk, int lddk, bf16_t* __restrict__ dv, int lddv,
                                                                    const float* __restrict__ bq, const float* __restrict__ bk,
                                                                    const float* __restrict__ bv) {
    extern __shared__ __attribute__((aligned(16))) unsigned char as_raw[];
    AsShared& sh = *reinterpret_cast<AsShared*>(as_raw);
    if (seed_dev) seed += *seed_dev;
    const int bh = blockIdx.x, b = bh / H, h = bh - b * H;
    as_load(sh.q, q, ldq, b, h, S, dh, bq);
    as_load(sh.k, kmat, ldk, b, h, S, dh, bk);
    as_load(sh.v, v, ldv, b, h, S, dh, bv);
    as_load(sh.g, dctx, lddo, b, h, S, dh, nullptr);
    __syncthreads();
    as_probabilities<false>(sh, key_pad, b, bh, S, dh, scale, drop_p, seed, const_cast<float*>(stats));
    const int quarter = dh >> 2;
    // dV[j] = sum_i pd[i][j] dO[i]: four features per thread
    for (int e = threadIdx.x; e < S * quarter; e += AS_THREADS) {
        const int j = e / quarter, c = (e - j * quarter) * 4;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int i = 0; i < S; ++i) {
            const float w = sh.pd[i][j];
            const float4 x = *reinterpret_cast<const float4*>(&sh.g[i][c]);
            a.x += w * x.x; a.y += w * x.y; a.z += w * x.z; a.w += w * x.w;
        }
        *reinterpret_cast<uint2*>(dv + ((size_t)b * S + j) * lddv + h * dh + c) = make_uint2(pack2bf(a.x, a.y), pack2bf(a.z, a.w));
    }
    __syncthreads();
    // dP (w.r.t. the dropped-out probabilities) masked and rescaled: keep <=> pd != 0 or p == 0
    const float dscale = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    for (int e = threadIdx.x; e < S * S; e += AS_THREADS) {
        const int i = e / S, j = e - i * S;
        const float acc = as_dot(sh.g[i], sh.v[j], dh);
        const bool keep = drop_p <= 0.f || sh.pd[i][j] != 0.f;
        sh.pd[i][j] = keep ? acc * dscale : 0.f;
    }
    __syncthreads();
    {
        const int G = S <= 16 ? 16 : (S <= 32 ? 32 : 64), jl = threadIdx.x & (G - 1);
        for (int i = threadIdx.x / G; i < S; i += AS_THREADS / G) {
            float t = jl < S ? sh.p[i][jl] * sh.pd[i][jl] : 0.f;
            for (int o = G >> 1; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
            if (jl == 0) sh.rowdot[i] = t;
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < S * S; e += AS_THREADS) {
        const int i = e / S, j = e - i * S;
        sh.p[i][j] = sh.p[i][j] * (sh.pd[i][j] - sh.rowdot[i]) * scale;      // dS, with the score scale folded in
    }
    __syncthreads();
    // dQ[i] = sum_j dS[i][j] k[j] (threads 0 .. S*dh/4) and dK[i] = sum_j dS[j][i] q[j] (the next S*dh/4): four features per thread
    for (int e = threadIdx.x; e < 2 * S * quarter; e += AS_THREADS) {
        const bool is_k = e >= S * quarter;
        const int r = is_k ? e - S * quarter : e;
        const int i = r / quarter, c = (r - i * quarter) * 4;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int j = 0; j < S; ++j) {
            const float w = is_k ? sh.p[j][i] : sh.p[i][j];
            const float4 x = *reinterpret_cast<const float4*>(is_k ? &sh.q[j][c] : &sh.k[j][c]);
            a.x += w * x.x; a.y += w * x.y; a.z += w * x.z; a.w += w * x.w;
        }
        bf16_t* const out = is_k ? dk + ((size_t)b * S + i) * lddk : dq + ((size_t)b * S + i) * lddq;
        *reinterpret_cast<uint2*>(out + h * dh + c) = make_uint2(pack2bf(a.x, a.y), pack2bf(a.z, a.w));
    }
}

}  // namespace toist

using namespace toist;

static int attn_small_ok(const char* who, int B, int H, int S, int dh, float drop_p) {
    TOIST_REQUIRE(B > 0 && H > 0 && S > 0 && S <= AS_MAXS && dh > 0 && dh <= AS_MAXD && (dh % 8) == 0, "%s: needs S <= %d, head dim <= %d and %% 8 == 0 (S=%d, dh=%d)",
                  who, AS_MAXS, AS_MAXD, S, dh);
    TOIST_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "%s: bad dropout p", who);
    return TOIST_OK;
}

static int attn_small_lds(const void* fn) {   // 118 KiB of LDS: opt in once, thread-safe
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(AsShared)) != hipSuccess) {
        set_last_error("toist_attn_small: cannot enable %zu bytes of LDS", sizeof(AsShared));
        return TOIST_EHIP;
    }
    return TOIST_OK;
}

extern "C" int toist_attn_small_fwd(const void* q, int ldq, const void* kmat, int ldk, const void* v, int ldv, const uint8_t* key_pad, int B, int H, int S,
                                    int dh, float scale, float drop_p, uint64_t seed, const uint64_t* seed_dev, void* ctx, int ldo, float* stats,
                                    const float* bq, const float* bk, const float* bv, void* stream) {
    if (int rc = attn_small_ok("toist_attn_small_fwd", B, H, S, dh, drop_p)) return rc;
    TOIST_REQUIRE(q && kmat && v && ctx && stats && (ldq % 8) == 0 && (ldk % 8) == 0 && (ldv % 8) == 0 && (ldo % 4) == 0, "toist_attn_small_fwd: bad args");
    static const int once = attn_small_lds((const void*)attn_small_fwd_kernel);
    if (once != TOIST_OK) return once;
    hipLaunchKernelGGL(attn_small_fwd_kernel, dim3(B * H), dim3(AS_THREADS), sizeof(AsShared), (hipStream_t)stream, (const bf16_t*)q, ldq, (const bf16_t*)kmat,
                       ldk, (const bf16_t*)v, ldv, key_pad, H, S, dh, scale, drop_p, (unsigned long long)seed, (const unsigned long long*)seed_dev,
                       (bf16_t*)ctx, ldo, stats, bq, bk, bv);
    return check_launch("toist_attn_small_fwd");
}

extern "C" int toist_attn_small_bwd(const void* q, int ldq, const void* kmat, int ldk, const void* v, int ldv, const uint8_t* key_pad, int B, int H, int S,
                                    int dh, float scale, float drop_p, uint64_t seed, const uint64_t* seed_dev, const float* stats, const void* dctx,
                                    int lddo, void* dq, int lddq, void* dk, int lddk, void* dv, int lddv, const float* bq, const float* bk,
                                    const float* bv, void* stream) {
    if (int rc = attn_small_ok("toist_attn_small_bwd", B, H, S, dh, drop_p)) return rc;
    TOIST_REQUIRE(q && kmat && v && stats && dctx && dq && dk && dv && (ldq % 8) == 0 && (ldk % 8) == 0 && (ldv % 8) == 0 && (lddo % 8) == 0 &&
                      (lddq % 4) == 0 && (lddk % 4) == 0 && (lddv % 4) == 0, "toist_attn_small_bwd: bad args");
    static const int once = attn_small_lds((const void*)attn_small_bwd_kernel);
    if (once != TOIST_OK) return once;
    hipLaunchKernelGGL(attn_small_bwd_kernel, dim3(B * H), dim3(AS_THREADS), sizeof(AsShared), (hipStream_t)stream, (const bf16_t*)q, ldq, (const bf16_t*)kmat,
                       ldk, (const bf16_t*)v, ldv, key_pad, H, S, dh, scale, drop_p, (unsigned long long)seed, (const unsigned long long*)seed_dev, stats,
                       (const bf16_t*)dctx, lddo, (bf16_t*)dq, lddq, (bf16_t*)dk, lddk, (bf16_t*)dv, lddv, bq, bk, bv);
    return check_launch("toist_attn_small_bwd");
}
