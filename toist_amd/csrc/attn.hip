// Fused attention core, forward: scores = scale * Q K^T -> key-padding mask -> softmax -> dropout -> P V in ONE launch
// (nn.MultiheadAttention's core, /root/reference/models/transformer.py:297,370-400 via torch; head dim 32 = d_model 256 / 8
// heads).  Replaces three launches (score GEMM, softmax, context GEMM) and the bf16 round trip of the scores; the
// probabilities (and their dropped-out copy) are still written once because the backward kernels consume them.
//
// One workgroup = 64 queries of one (image, head); a wavefront owns 16 of them.  K and V of the head ([Sk, 32] each) are
// staged in LDS once.  Scores come out of the MFMA with lane (query c16, group g) holding keys 16j + 4g .. +3 of every
// 16-key block j, i.e. a whole score row is spread over just four lanes: max / sum need two shuffles.  For P V the MFMA
// reduction slots are assigned to keys in exactly that order (slot 8g + r <-> key 32c + 4g + r, slot 8g + 4 + r <-> key
// 32c + 16 + 4g + r), so the probabilities are used as the A operand straight from their registers and V is read
// k-major from LDS (ds_read_b64_tr_b16) at the matching key offsets -- no transposition of P anywhere.
#include "common.h"

namespace toist {

template <int NB>   // 16-key blocks (Sk <= 16 * NB), NB even
__global__ __launch_bounds__(256) void attn_fwd_kernel(const bf16_t* __restrict__ q, int ldq, const bf16_t* __restrict__ kmat, int ldk,
                                                       const bf16_t* __restrict__ v, int ldv, const unsigned char* __restrict__ key_pad, int H,
                                                       int Sq, int Sk, int ld, float scale, bf16_t* __restrict__ prob, bf16_t* __restrict__ prob_drop,
                                                       float drop_p, unsigned long long seed, const unsigned long long* __restrict__ seed_dev,
                                                       bf16_t* __restrict__ ctx, int ldo) {
    constexpr int SKP = NB * 16, DH = 32;
    extern __shared__ __attribute__((aligned(16))) bf16_t smem[];
    bf16_t* sK = smem;               // [SKP][32], 16-byte chunk (key, c) stored in slot c ^ ((key >> 1) & 3)
    bf16_t* sV = smem + SKP * DH;    // [SKP][32] plain (read k-major)
    unsigned char* sDead = reinterpret_cast<unsigned char*>(smem + 2 * SKP * DH);   // [SKP] 1 = key masked (padding or beyond Sk)
    typedef __attribute__((address_space(3))) s16x4_t* lds_v4;
    if (seed_dev) seed += *seed_dev;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c16 = lane & 15;
    const int bh = blockIdx.y, b = bh / H, h = bh - b * H;

    // ---- K, V of this head -> LDS (rows beyond Sk are zero) ----
    for (int c = tid; c < SKP * 4; c += 256) {
        const int key = c >> 2, ch = c & 3;
        uint4 kv = make_uint4(0, 0, 0, 0), vv = make_uint4(0, 0, 0, 0);
        if (key < Sk) {
            kv = *reinterpret_cast<const uint4*>(kmat + ((size_t)b * Sk + key) * ldk + h * DH + ch * 8);
            vv = *reinterpret_cast<const uint4*>(v + ((size_t)b * Sk + key) * ldv + h * DH + ch * 8);
        }
        *reinterpret_cast<uint4*>(sK + key * DH + ((ch ^ ((key >> 1) & 3)) << 3)) = kv;
        *reinterpret_cast<uint4*>(sV + key * DH + (ch << 3)) = vv;
    }
    for (int kk = tid; kk < SKP; kk += 256) sDead[kk] = (kk >= Sk || (key_pad != nullptr && key_pad[(size_t)b * Sk + kk])) ? 1 : 0;
    __syncthreads();

    const int qi = blockIdx.x * 64 + wave * 16 + c16;           // this lane's query
    const bool qlive = qi < Sq;
    bf16x8_t qf = {0, 0, 0, 0, 0, 0, 0, 0};
    if (qlive) qf = *reinterpret_cast<const bf16x8_t*>(q + ((size_t)b * Sq + qi) * ldq + h * DH + g * 8);

    // ---- scores: lane (query c16, group g) gets keys 16 j + 4 g + r ----
    f32x4_t s[NB];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int key = j * 16 + c16;                           // B-operand row of this lane
        const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(sK + key * DH + ((g ^ ((key >> 1) & 3)) << 3));
        f32x4_t a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf, f32x4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        const unsigned dead4 = *reinterpret_cast<const unsigned*>(sDead + j * 16 + g * 4);   // this lane's 4 keys
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            a[r] = ((dead4 >> (8 * r)) & 0xffu) ? -INFINITY : a[r] * scale;
            mx = fmaxf(mx, a[r]);
        }
        s[j] = a;
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) { s[j][r] = __expf(s[j][r] - mx); sum += s[j][r]; }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.f / sum;
    const unsigned thresh = prob_drop ? (unsigned)(drop_p * 4294967296.0) : 0u;
    const float dscale = prob_drop ? 1.f / (1.f - drop_p) : 1.f;
    const size_t row = (size_t)bh * Sq + (qlive ? qi : 0);
    unsigned pk[NB][2];                                          // bf16 pairs of the probabilities that enter P V
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        float o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = s[j][r] * inv;
        const int k0 = j * 16 + g * 4;
        const bool inrow = qlive && k0 < ld;
        if (inrow) *reinterpret_cast<uint2*>(prob + row * ld + k0) = make_uint2(pack2bf(o[0], o[1]), pack2bf(o[2], o[3]));
        if (prob_drop) {
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = dropout_keep(seed, (unsigned long long)row * ld + k0 + r, thresh) ? o[r] * dscale : 0.f;
            if (inrow) *reinterpret_cast<uint2*>(prob_drop + row * ld + k0) = make_uint2(pack2bf(o[0], o[1]), pack2bf(o[2], o[3]));
        }
        pk[j][0] = pack2bf(o[0], o[1]);
        pk[j][1] = pack2bf(o[2], o[3]);
    }

    // ---- context = P V: slot 8g + r <-> key 32c + 4g + r, slot 8g + 4 + r <-> key 32c + 16 + 4g + r ----
    f32x4_t acc[2] = {f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int c = 0; c < NB / 2; ++c) {
        union { unsigned u[4]; bf16x8_t v8; } pa;
        pa.u[0] = pk[2 * c][0]; pa.u[1] = pk[2 * c][1]; pa.u[2] = pk[2 * c + 1][0]; pa.u[3] = pk[2 * c + 1][1];
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            const int col = nb * 16 + ((c16 & 3) >> 1) * 8 + (c16 & 1) * 4;
            const int k_lo = 32 * c + 4 * g + (c16 >> 2), k_hi = k_lo + 16;
            union { struct { s16x4_t a, b; } hh; bf16x8_t v8; } vb;
            vb.hh.a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(sV + k_lo * DH + col));
            vb.hh.b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(sV + k_hi * DH + col));
            acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vb.v8, pa.v8, acc[nb], 0, 0, 0);
        }
    }
    if (qlive) {
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
            *reinterpret_cast<uint2*>(ctx + ((size_t)b * Sq + qi) * ldo + h * DH + nb * 16 + g * 4) =
                make_uint2(pack2bf(acc[nb][0], acc[nb][1]), pack2bf(acc[nb][2], acc[nb][3]));
    }
}

}  // namespace toist

using namespace toist;

extern "C" int toist_attn_fwd(const void* q, int ldq, const void* kmat, int ldk, const void* v, int ldv, const uint8_t* key_pad, int B, int H, int Sq,
                              int Sk, int dh, int ld, float scale, void* prob, void* prob_drop, float drop_p, uint64_t seed, const uint64_t* seed_dev,
                              void* ctx, int ldo, void* stream) {
    TOIST_REQUIRE(q && kmat && v && prob && ctx && B > 0 && H > 0 && Sq > 0 && Sk > 0, "toist_attn_fwd: bad args");
    TOIST_REQUIRE(dh == 32, "toist_attn_fwd: head dim must be 32 (got %d)", dh);
    TOIST_REQUIRE(Sk <= 480 && ld >= Sk && (ld % 8) == 0, "toist_attn_fwd: Sk <= 480 (K, V and the mask of a head must fit 64 KB of LDS) and ld = round8(Sk) (got %d, %d)", Sk, ld);
    TOIST_REQUIRE((ldq % 8) == 0 && (ldk % 8) == 0 && (ldv % 8) == 0 && (ldo % 4) == 0, "toist_attn_fwd: row strides must keep 16-byte alignment");
    TOIST_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "toist_attn_fwd: bad dropout p");
    const dim3 grid((Sq + 63) / 64, B * H), block(256);
    hipStream_t st = (hipStream_t)stream;
#define TOIST_ATTN(NB)                                                                                                                       \
    hipLaunchKernelGGL((attn_fwd_kernel<NB>), grid, block, 2 * (NB) * 16 * 32 * sizeof(bf16_t) + (NB) * 16, st, (const bf16_t*)q, ldq, (const bf16_t*)kmat, ldk, \
                       (const bf16_t*)v, ldv, key_pad, H, Sq, Sk, ld, scale, (bf16_t*)prob, (bf16_t*)prob_drop, drop_p,                      \
                       (unsigned long long)seed, (const unsigned long long*)seed_dev, (bf16_t*)ctx, ldo)
    if (Sk <= 128) TOIST_ATTN(8);
    else if (Sk <= 416) TOIST_ATTN(26);
    else TOIST_ATTN(30);
#undef TOIST_ATTN
    return check_launch("toist_attn_fwd");
}
