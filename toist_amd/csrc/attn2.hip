// Attention cores, second generation (head dim 32 = d_model 256 / 8 heads): nn.MultiheadAttention's core as called from
// /root/reference/models/transformer.py:297 (encoder self-attention) and :370-400 (decoder self- / cross-attention).
//
// Forward (attn2_fwd_kernel): scores -> key-padding mask -> softmax -> dropout -> P V in one launch, flash style -- the keys are
// walked in blocks of 128 with a running row maximum / row sum, so no score-shaped tensor exists anywhere and the key count is
// unbounded (K / V are staged in LDS 512 keys at a time; the reference validates at 800 x 1333 pixels = ~1100 tokens).  Per score
// element the vector pipe issues: max, subtract + multiply + exp2, add, half a dropout hash (one 32-bit hash decides
// TWO keys, 16 bits each), compare + select, half a bf16 pack -- ~12 operations against ~30 in the first-generation kernel
// (csrc/attn.hip); the mask path runs only for 128-key blocks that contain a masked key.  The 1 / row sum and 1 / (1 - p) factors are
// applied to the 32 outputs of a query, not to its scores.
//
// Backward (attn2_bwd_kernel): KEY-OWNING.  A wavefront owns 32 keys of one (image, head) -- K, V and K^T fragments stay in its
// registers for the whole launch -- and walks all queries in tiles of 32.  The score-shaped quantities come out of the MFMA with
// lane = key, four queries per 16 x 16 block, which is exactly the operand layout dK^T += Q^T dS and dV^T += dO^T Pd want: dK and dV
// accumulate in registers and are written ONCE (the first-generation kernel split the queries over workgroups and left 4 partial
// copies of dK / dV, 14.5 MB per encoder layer, to a fold kernel).  dQ needs the other orientation: dS goes through a wave-private
// LDS tile (no barrier: a wave's LDS operations execute in order), comes back transposed (ds_read_b64_tr_b16) and the wave's share
// K^T dS^T of dQ^T is summed over the four waves of the workgroup in LDS -- the only barrier of a query tile.  Workgroups of one
// head own different key ranges, so dQ leaves as ONE partial per workgroup (bf16 [splits][B * Sq][256]); the consumer -- the
// in_proj data-gradient launch of csrc/tlayer.hip -- adds the partials while it stages its A rows.  P is re-formed from the forward's
// (row maximum, 1 / row sum): P = exp2((s - m) c) / sum, the expression of the forward pass, and the keep mask from the same hash.
#include "common.h"

#include <type_traits>

namespace toist {

namespace {

typedef __attribute__((ext_vector_type(8))) short s16x8_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
typedef __attribute__((address_space(3))) s16x4_t* lds_v4;

constexpr float LOG2E = 1.4426950408889634f;

// One 32-bit hash decides two adjacent keys of a score row: element (row, key) -> pair (row * ld + key) >> 1, field key & 1 (bits 0-15 /
// 16-31), kept when field >= round(p * 65536).  Two multiply rounds on the full-rate 24-bit multiplier (v_mad_u32_u24; a 32-bit
// v_mul_lo_u32 occupies the vector pipe four times as long); keep rate, field / neighbour / stride correlations measured in
// tools/r4/hash_quality.py.
__device__ __forceinline__ unsigned pair_hash(unsigned pair, unsigned s0, unsigned s1) {
    unsigned a = pair ^ s0;
    a ^= a >> 12;               // the 24-bit multiply below only sees bits 0-23: fold the upper bits in first (pairs 2^24 apart otherwise share 99.9 % of their masks)
    unsigned h = __umul24(a, 0x9E3779u) + s1;
    h ^= h >> 15;
    h = __umul24(h, 0x85EBCBu) + (a >> 8);
    h ^= h >> 13;
    return h;
}

__device__ __forceinline__ bf16x8_t tr_pair(const bf16_t* lo_ptr, const bf16_t* hi_ptr) {
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)lo_ptr);
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)hi_ptr);
    return __builtin_bit_cast(bf16x8_t, (s16x8_t)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

__device__ __forceinline__ bf16x8_t frag_of(unsigned a, unsigned b, unsigned c, unsigned d) {
    const u32x4_t u = {a, b, c, d};
    return __builtin_bit_cast(bf16x8_t, u);
}

template <int N, typename F>
__device__ __forceinline__ void unrolled(F&& f) {
    if constexpr (N > 0) {
        unrolled<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

}  // namespace

constexpr int A2_SKB = 512;     // keys staged in LDS at a time (forward)

__global__ __launch_bounds__(256) void attn2_fwd_kernel(const bf16_t* __restrict__ q, int ldq, const bf16_t* __restrict__ kmat, int ldk,
                                                        const bf16_t* __restrict__ v, int ldv, const unsigned char* __restrict__ key_pad, int H,
                                                        int Sq, int Sk, int ldp, float scale, float drop_p, unsigned long long seed,
                                                        const unsigned long long* __restrict__ seed_dev, bf16_t* __restrict__ ctx, int ldo,
                                                        float* __restrict__ lse) {
    constexpr int DH = 32;
    extern __shared__ __attribute__((aligned(16))) bf16_t smem[];
    bf16_t* sK = smem;                        // [512][32], 16-byte chunk (key, c) stored in slot c ^ ((key >> 1) & 3)
    bf16_t* sV = smem + A2_SKB * DH;          // [512][32] plain (read k-major)
    unsigned char* sDead = reinterpret_cast<unsigned char*>(smem + 2 * A2_SKB * DH);   // [512] 1 = key masked (padding or beyond Sk)
    if (seed_dev) seed += *seed_dev;
    const unsigned s0 = (unsigned)seed, s1 = (unsigned)(seed >> 32) ^ ((unsigned)seed * 0x9E3779B9u);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c16 = lane & 15;
    const int bh = blockIdx.y, b = bh / H, h = bh - b * H;
    const int qi = blockIdx.x * 64 + wave * 16 + c16;           // this lane's query
    const bool qlive = qi < Sq;
    // exp2((s - m) * c), c = scale * log2(e): the subtraction comes FIRST and is exact for the row maximum.  (s * c - m * c as one fma is
    // wrong for the scores of an undamped random-init backbone: at |s c| ~ 1e11 the rounded product m * c is off by thousands and exp2
    // overflows -- found as NaN in the second training step of the bench model; folding c into a re-rounded bf16 q costs half a digit.)
    const float c = scale * LOG2E;
    bf16x8_t qf = {0, 0, 0, 0, 0, 0, 0, 0};
    if (qlive) qf = *reinterpret_cast<const bf16x8_t*>(q + ((size_t)b * Sq + qi) * ldq + h * DH + g * 8);
    const bool dropping = drop_p > 0.f;
    const unsigned t16 = dropping ? (unsigned)(drop_p * 65536.0f + 0.5f) : 0u;
    const unsigned row = (unsigned)bh * (unsigned)Sq + (unsigned)(qlive ? qi : 0);
    const unsigned pair_row = row * (unsigned)(ldp >> 1);

    float m = -INFINITY, l = 0.f;
    f32x4_t acc[2] = {f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}};

    for (int k0 = 0; k0 < Sk; k0 += A2_SKB) {
        const int left = Sk - k0;
        const int nkeys = left < A2_SKB ? ((left + 127) & ~127) : A2_SKB;      // rows staged: whole 128-key blocks
        if (k0 > 0) __syncthreads();                                          // every wave has consumed the previous 512 keys
        // 16-byte pieces of the K / V rows: eight requests in flight per thread before the first LDS store (a load -> store -> load chain of
        // eight round trips was half of the launch time at 416 keys)
        for (int base = 0; base < nkeys * 4; base += 1024) {
            u32x4_t kv[4], vv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int cch = base + u * 256 + tid, kl = cch >> 2, ch = cch & 3, key = k0 + kl;
                kv[u] = u32x4_t{0, 0, 0, 0};
                vv[u] = u32x4_t{0, 0, 0, 0};
                if (cch < nkeys * 4 && key < Sk) {
                    kv[u] = *reinterpret_cast<const u32x4_t*>(kmat + ((size_t)b * Sk + key) * ldk + h * DH + ch * 8);
                    vv[u] = *reinterpret_cast<const u32x4_t*>(v + ((size_t)b * Sk + key) * ldv + h * DH + ch * 8);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int cch = base + u * 256 + tid, kl = cch >> 2, ch = cch & 3;
                if (cch < nkeys * 4) {
                    *reinterpret_cast<u32x4_t*>(sK + kl * DH + ((ch ^ ((kl >> 1) & 3)) << 3)) = kv[u];
                    *reinterpret_cast<u32x4_t*>(sV + kl * DH + (ch << 3)) = vv[u];
                }
            }
        }
        for (int kl = tid; kl < nkeys; kl += 256) sDead[kl] = (k0 + kl >= Sk || (key_pad != nullptr && key_pad[(size_t)b * Sk + k0 + kl])) ? 1 : 0;
        __syncthreads();

        for (int kb = 0; kb < nkeys / 128; ++kb) {
            const int kbase = kb * 128;
            // ---- raw dot products: lane (query c16, group g) gets keys kbase + 16 j + 4 g + r ----
            f32x4_t s[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int kl = kbase + j * 16 + c16;
                const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(sK + kl * DH + ((g ^ ((kl >> 1) & 3)) << 3));
                s[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf, f32x4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            }
            // the mask path only where this block holds a masked key (wave-uniform)
            const unsigned short dd = *reinterpret_cast<const unsigned short*>(sDead + kbase + lane * 2);
            if (__any(dd != 0)) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const unsigned dead4 = *reinterpret_cast<const unsigned*>(sDead + kbase + j * 16 + g * 4);
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if ((dead4 >> (8 * r)) & 0xffu) s[j][r] = -INFINITY;
                }
            }
            float bm = -INFINITY;
#pragma unroll
            for (int j = 0; j < 8; ++j) bm = fmaxf(fmaxf(bm, fmaxf(s[j][0], s[j][1])), fmaxf(s[j][2], s[j][3]));
            bm = fmaxf(bm, __shfl_xor(bm, 16, 64));
            bm = fmaxf(bm, __shfl_xor(bm, 32, 64));
            const float mn = fmaxf(m, bm);
            const float msafe = (mn == -INFINITY) ? 0.f : mn;            // a row with nothing but masked keys so far
            const float alpha = __builtin_amdgcn_exp2f((m - msafe) * c);   // m = -inf: 0
            m = mn;
            float psum = 0.f;
            unsigned pk[8][2];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float p[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    p[r] = __builtin_amdgcn_exp2f((s[j][r] - msafe) * c);
                    psum += p[r];
                }
                if (dropping) {
                    const unsigned pair = pair_row + (unsigned)((k0 + kbase + j * 16 + 4 * g) >> 1);
                    const unsigned h0 = pair_hash(pair, s0, s1), h1 = pair_hash(pair + 1u, s0, s1);
                    p[0] = (h0 & 0xffffu) >= t16 ? p[0] : 0.f;
                    p[1] = (h0 >> 16) >= t16 ? p[1] : 0.f;
                    p[2] = (h1 & 0xffffu) >= t16 ? p[2] : 0.f;
                    p[3] = (h1 >> 16) >= t16 ? p[3] : 0.f;
                }
                pk[j][0] = pack2bf(p[0], p[1]);
                pk[j][1] = pack2bf(p[2], p[3]);
            }
            l = l * alpha + psum;
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[nb][r] *= alpha;
            // ---- context += P V: slot 8g + r <-> key 32c + 4g + r, slot 8g + 4 + r <-> key 32c + 16 + 4g + r ----
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                const bf16x8_t pa = frag_of(pk[2 * cc][0], pk[2 * cc][1], pk[2 * cc + 1][0], pk[2 * cc + 1][1]);
                const int k_lo = kbase + 32 * cc + 4 * g + (c16 >> 2), k_hi = k_lo + 16;
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) {
                    const int col = nb * 16 + (c16 & 3) * 4;
                    const bf16x8_t vb = tr_pair(sV + k_lo * DH + col, sV + k_hi * DH + col);
                    acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vb, pa, acc[nb], 0, 0, 0);
                }
            }
        }
    }
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.f / l;
    const float os = dropping ? inv / (1.f - drop_p) : inv;
    if (qlive) {
        if (g == 0) *reinterpret_cast<float2*>(lse + 2 * (size_t)row) = make_float2(m, inv);     // (row maximum of the RAW dot products, 1 / row sum)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
            *reinterpret_cast<uint2*>(ctx + ((size_t)b * Sq + qi) * ldo + h * DH + nb * 16 + g * 4) =
                make_uint2(pack2bf(acc[nb][0] * os, acc[nb][1] * os), pack2bf(acc[nb][2] * os, acc[nb][3] * os));
    }
}

// ---- backward ----------------------------------------------------------------------------------------------------------------------
// The body lives in attn2_bwd_body.h (shared with the XCD-resident decoder backward of csrc/xdec.hip).
#include "attn2_bwd_body.h"
using a2b::A2_BWD_LDS;
using a2b::A2_NG;

__global__ __launch_bounds__(256 * A2_NG) void attn2_bwd_kernel(const bf16_t* __restrict__ q, int ldq, const bf16_t* __restrict__ kmat, int ldk,
                                                        const bf16_t* __restrict__ v, int ldv, const bf16_t* __restrict__ ctx, int ldo,
                                                        const bf16_t* __restrict__ dctx, int lddo, const float* __restrict__ lse,
                                                        const unsigned char* __restrict__ key_pad, int H, int Sq, int Sk, int ldp, float scale,
                                                        float drop_p, unsigned long long seed, const unsigned long long* __restrict__ seed_dev,
                                                        bf16_t* __restrict__ dq, int lddq, bf16_t* __restrict__ dk, int lddk,
                                                        bf16_t* __restrict__ dv, int lddv, bf16_t* __restrict__ dq_part, long long part_stride) {
    extern __shared__ __attribute__((aligned(16))) unsigned char a2_smem[];
    a2b::attn2_bwd_body<false>(a2_smem, (int)blockIdx.x, (int)blockIdx.y, q, ldq, kmat, ldk, v, ldv, ctx, ldo, dctx, lddo, lse, key_pad, H, Sq, Sk, ldp, scale, drop_p, seed,
                               seed_dev, dq, lddq, dk, lddk, dv, lddv, dq_part, part_stride);
}

}  // namespace toist

using namespace toist;

extern "C" int toist_attn2_splits(int Sk) { return Sk > 0 ? ((Sk + 31) / 32 + 3) / 4 : 0; }

extern "C" int toist_attn2_fwd(const void* q, int ldq, const void* kmat, int ldk, const void* v, int ldv, const uint8_t* key_pad, int B, int H, int Sq,
                               int Sk, int dh, float scale, float drop_p, uint64_t seed, const uint64_t* seed_dev, void* ctx, int ldo, float* lse,
                               void* stream) {
    TOIST_REQUIRE(q && kmat && v && ctx && lse && B > 0 && H > 0 && Sq > 0 && Sk > 0, "toist_attn2_fwd: bad args");
    TOIST_REQUIRE(dh == 32, "toist_attn2_fwd: head dim must be 32 (got %d)", dh);
    TOIST_REQUIRE((ldq % 8) == 0 && (ldk % 8) == 0 && (ldv % 8) == 0 && (ldo % 4) == 0, "toist_attn2_fwd: row strides must keep 16-byte (q, k, v) / 8-byte (ctx) alignment");
    TOIST_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "toist_attn2_fwd: bad dropout p");
    const int ldp = (Sk + 7) / 8 * 8;
    TOIST_REQUIRE((long long)B * H * Sq * ldp < (1ll << 32), "toist_attn2_fwd: B * H * Sq * round8(Sk) must stay below 2^32 (dropout element index)");
    const size_t lds = (size_t)2 * A2_SKB * 32 * sizeof(bf16_t) + A2_SKB;
    static std::atomic<unsigned long long> done{0};
    if (!lds_attr_once_flag(done, [] { return hipFuncSetAttribute((const void*)attn2_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess; })) {
        set_last_error("toist_attn2_fwd: cannot raise the dynamic LDS limit");
        return TOIST_EHIP;
    }
    hipLaunchKernelGGL(attn2_fwd_kernel, dim3((Sq + 63) / 64, B * H), dim3(256), lds, (hipStream_t)stream, (const bf16_t*)q, ldq, (const bf16_t*)kmat, ldk,
                       (const bf16_t*)v, ldv, key_pad, H, Sq, Sk, ldp, scale, drop_p, (unsigned long long)seed, (const unsigned long long*)seed_dev,
                       (bf16_t*)ctx, ldo, lse);
    return check_launch("toist_attn2_fwd");
}

extern "C" int toist_attn2_bwd(const void* q, int ldq, const void* kmat, int ldk, const void* v, int ldv, const void* ctx, int ldo, const void* dctx, int lddo,
                               const float* lse, const uint8_t* key_pad, int B, int H, int Sq, int Sk, int dh, float scale, float drop_p, uint64_t seed,
                               const uint64_t* seed_dev, void* dq, int lddq, void* dk, int lddk, void* dv, int lddv, void* dq_part, void* stream) {
    TOIST_REQUIRE(q && kmat && v && ctx && dctx && lse && dk && dv && B > 0 && H > 0 && Sq > 0 && Sk > 0, "toist_attn2_bwd: bad args");
    TOIST_REQUIRE(dh == 32, "toist_attn2_bwd: head dim must be 32 (got %d)", dh);
    const int splits = toist_attn2_splits(Sk);
    TOIST_REQUIRE(splits == 1 ? dq != nullptr : dq_part != nullptr,
                  "toist_attn2_bwd: %d key splits: dq for one, dq_part (bf16 [splits][B*Sq][H*32], folded by the consumer) for more", splits);
    TOIST_REQUIRE((ldq % 8) == 0 && (ldk % 8) == 0 && (ldv % 8) == 0 && (ldo % 4) == 0 && (lddo % 8) == 0 && (lddq % 4) == 0 && (lddk % 4) == 0 && (lddv % 4) == 0,
                  "toist_attn2_bwd: row strides must keep 16-byte (q, k, v, dctx) / 8-byte (ctx, gradients) alignment");
    TOIST_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "toist_attn2_bwd: bad dropout p");
    const int ldp = (Sk + 7) / 8 * 8;
    TOIST_REQUIRE((long long)B * H * Sq * ldp < (1ll << 32), "toist_attn2_bwd: B * H * Sq * round8(Sk) must stay below 2^32 (dropout element index)");
    static std::atomic<unsigned long long> done{0};
    if (!lds_attr_once_flag(done, [] { return hipFuncSetAttribute((const void*)attn2_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess; })) {
        set_last_error("toist_attn2_bwd: cannot raise the dynamic LDS limit");
        return TOIST_EHIP;
    }
    hipLaunchKernelGGL(attn2_bwd_kernel, dim3(splits, B * H), dim3(256 * A2_NG), A2_BWD_LDS, (hipStream_t)stream, (const bf16_t*)q, ldq, (const bf16_t*)kmat, ldk,
                       (const bf16_t*)v, ldv, (const bf16_t*)ctx, ldo, (const bf16_t*)dctx, lddo, lse, key_pad, H, Sq, Sk, ldp, scale, drop_p,
                       (unsigned long long)seed, (const unsigned long long*)seed_dev, (bf16_t*)dq, lddq, (bf16_t*)dk, lddk, (bf16_t*)dv, lddv,
                       splits > 1 ? (bf16_t*)dq_part : (bf16_t*)nullptr, (long long)B * Sq * H * 32);
    return check_launch("toist_attn2_bwd");
}
