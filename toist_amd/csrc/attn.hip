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

// s * scale - m with the product rounded to fp32 before the subtraction (no fma contraction): the forward kernel takes the row
// maximum of the ROUNDED products, so only this order re-forms exp(s - max) bit for bit; with scores of 1e8 and more (an undamped
// random-init backbone) a fused multiply-subtract is off by the product's rounding error, i.e. by tens, before the exp
__device__ __forceinline__ float scaled_minus(float s, float scale, float m) {
#pragma clang fp contract(off)
    const float p = s * scale;
    return p - m;
}

template <int NB>   // 16-key blocks (Sk <= 16 * NB), NB even
__global__ __launch_bounds__(256) void attn_fwd_kernel(const bf16_t* __restrict__ q, int ldq, const bf16_t* __restrict__ kmat, int ldk,
                                                       const bf16_t* __restrict__ v, int ldv, const unsigned char* __restrict__ key_pad, int H,
                                                       int Sq, int Sk, int ld, float scale, bf16_t* __restrict__ prob, bf16_t* __restrict__ prob_drop,
                                                       float drop_p, unsigned long long seed, const unsigned long long* __restrict__ seed_dev,
                                                       bf16_t* __restrict__ ctx, int ldo, float* __restrict__ lse) {
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
            a[r] = ((dead4 >> (8 * r)) & 0xffu) ? -INFINITY : scaled_minus(a[r], scale, 0.f);
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
    const bool dropping = drop_p > 0.f;
    const unsigned thresh = dropping ? (unsigned)(drop_p * 4294967296.0) : 0u;
    const float dscale = dropping ? 1.f / (1.f - drop_p) : 1.f;
    const size_t row = (size_t)bh * Sq + (qlive ? qi : 0);
    // flash-style bookkeeping: with `lse` the backward kernel re-forms P = exp(scale q.k - lse) (and the dropout mask from the
    // same hash) instead of reading two [Sq, Sk] bf16 matrices per head back from HBM; prob / prob_drop may then be NULL
    // (row maximum, 1 / row sum) rather than one fp32 log-sum-exp: exp(s - max) * rsum re-forms P exactly as it was formed here, whereas
    // s - lse loses the low bits of s once |s| is large (at |s| ~ 1e8 the fp32 spacing is 8: exp() of that error is unbounded)
    if (lse != nullptr && qlive && g == 0) *reinterpret_cast<float2*>(lse + 2 * row) = make_float2(mx, inv);
    unsigned pk[NB][2];                                          // bf16 pairs of the probabilities that enter P V
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        float o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = s[j][r] * inv;
        const int k0 = j * 16 + g * 4;
        const bool inrow = qlive && k0 < ld;
        if (inrow && prob != nullptr) *reinterpret_cast<uint2*>(prob + row * ld + k0) = make_uint2(pack2bf(o[0], o[1]), pack2bf(o[2], o[3]));
        if (dropping) {
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = dropout_keep(seed, (unsigned long long)row * ld + k0 + r, thresh) ? o[r] * dscale : 0.f;
            if (inrow && prob_drop != nullptr) *reinterpret_cast<uint2*>(prob_drop + row * ld + k0) = make_uint2(pack2bf(o[0], o[1]), pack2bf(o[2], o[3]));
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

// Fused attention core, backward: from the saved probabilities, dO and the forward context it produces dQ, dK, dV of one
// (image, head) in ONE launch -- replaces dV = Pd^T dO, dPd = dO V^T, the softmax backward, dQ = scale dS K and
// dK = scale dS^T Q (four batched GEMMs + one row kernel, each launch-latency-bound at these sizes).
//
// One workgroup of 8 wavefronts per (image, head); queries in tiles of 64.
//   phase A (wave = a slice of NBW 16-key blocks): dPd = dO V^T comes out of the MFMA with lane (key c16, group g) holding
//     queries 4g .. 4g+3 of each 16-query block, so dS = P o (dP - D), D = rowsum(dO o O) (no cross-lane reduction), and
//     dS^T / Pd^T feed dK += dS^T Q and dV += Pd^T dO straight from registers as the B operand (reduction slot 8g' + r <->
//     query 32c + 4g' + r, slot 8g' + 4 + r <-> query 32c + 16 + 4g' + r); dK, dV accumulate in registers over all tiles.
//     dS is also written to LDS key-major.
//   phase B (wave = one 16 x 16 block of the tile's dQ): dQ = dS K with both operands read k-major from LDS.
template <int NBW>   // 16-key blocks per wave: keys <= 8 * NBW * 16
__global__ __launch_bounds__(512) void attn_bwd_kernel(const bf16_t* __restrict__ q, int ldq, const bf16_t* __restrict__ kmat, int ldk,
                                                       const bf16_t* __restrict__ v, int ldv, const bf16_t* __restrict__ prob,
                                                       const bf16_t* __restrict__ prob_drop, const bf16_t* __restrict__ ctx, int ldo,
                                                       const bf16_t* __restrict__ dctx, int lddo, int H, int Sq, int Sk, int ld, float scale,
                                                       float drop_p, bf16_t* __restrict__ dq, int lddq, bf16_t* __restrict__ dk, int lddk,
                                                       bf16_t* __restrict__ dv, int lddv) {
    constexpr int SKP = 8 * NBW * 16, DH = 32, QT = 64, TS = QT + 8;   // TS: row stride of the transposed tiles (bank spread)
    extern __shared__ __attribute__((aligned(16))) bf16_t smem[];
    bf16_t* sV = smem;                       // [SKP][32]
    bf16_t* sK = sV + SKP * DH;              // [SKP][32]
    bf16_t* sdST = sK + SKP * DH;            // [SKP][TS]   dS^T of the current tile
    bf16_t* sdO = sdST + SKP * TS;           // [QT][32]
    bf16_t* sdOT = sdO + QT * DH;            // [32][TS]
    bf16_t* sQT = sdOT + DH * TS;            // [32][TS]
    float* sD = reinterpret_cast<float*>(sQT + DH * TS);   // [QT]
    typedef __attribute__((address_space(3))) s16x4_t* lds_v4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c16 = lane & 15;
    const int bh = blockIdx.x, b = bh / H, h = bh - b * H;
    const float dscale = prob_drop ? 1.f / (1.f - drop_p) : 1.f;

    for (int c = tid; c < SKP * 4; c += 512) {
        const int key = c >> 2, ch = c & 3;
        uint4 kv = make_uint4(0, 0, 0, 0), vv = make_uint4(0, 0, 0, 0);
        if (key < Sk) {
            kv = *reinterpret_cast<const uint4*>(kmat + ((size_t)b * Sk + key) * ldk + h * DH + ch * 8);
            vv = *reinterpret_cast<const uint4*>(v + ((size_t)b * Sk + key) * ldv + h * DH + ch * 8);
        }
        *reinterpret_cast<uint4*>(sK + key * DH + (ch << 3)) = kv;
        *reinterpret_cast<uint4*>(sV + key * DH + (ch << 3)) = vv;
    }

    f32x4_t accK[NBW][2], accV[NBW][2];
#pragma unroll
    for (int kb = 0; kb < NBW; ++kb)
#pragma unroll
        for (int eb = 0; eb < 2; ++eb) { accK[kb][eb] = f32x4_t{0.f, 0.f, 0.f, 0.f}; accV[kb][eb] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }

    // The probabilities a lane needs are 2-byte elements of 4 different rows: un-vectorisable, and a global round trip per
    // key block would be fully exposed with two waves per SIMD.  They are fetched one key block ahead (across tile boundaries).
    unsigned short pnx[4][4], dnx[4][4];
    auto fetch = [&](int q0_, int kb_) {
        const int key = (wave * NBW + kb_) * 16 + c16;
#pragma unroll
        for (int xb = 0; xb < 4; ++xb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int qi = q0_ + xb * 16 + 4 * g + r;
                unsigned short pv = 0, pd = 0;
                if (key < Sk && qi < Sq) {
                    const size_t at = ((size_t)bh * Sq + qi) * ld + key;
                    pv = prob[at];
                    pd = prob_drop ? prob_drop[at] : pv;
                }
                pnx[xb][r] = pv;
                dnx[xb][r] = pd;
            }
    };
    fetch(0, 0);

    const int n_tiles = (Sq + QT - 1) / QT;
    for (int t = 0; t < n_tiles; ++t) {
        const int q0 = t * QT;
        {   // ---- tile prologue: dO, Q (transposed copies) and D = rowsum(dO o O) ----
            const int qq = tid >> 3, ch = tid & 7, qi = q0 + qq;             // 64 queries x 8 chunks of 4 features
            uint2 d2 = make_uint2(0, 0), o2 = make_uint2(0, 0), q2 = make_uint2(0, 0);
            if (qi < Sq) {
                d2 = *reinterpret_cast<const uint2*>(dctx + ((size_t)b * Sq + qi) * lddo + h * DH + ch * 4);
                o2 = *reinterpret_cast<const uint2*>(ctx + ((size_t)b * Sq + qi) * ldo + h * DH + ch * 4);
                q2 = *reinterpret_cast<const uint2*>(q + ((size_t)b * Sq + qi) * ldq + h * DH + ch * 4);
            }
            const bf16_t* dp = reinterpret_cast<const bf16_t*>(&d2);
            const bf16_t* op = reinterpret_cast<const bf16_t*>(&o2);
            const bf16_t* qp = reinterpret_cast<const bf16_t*>(&q2);
            float part = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                part += bf2f(dp[j]) * bf2f(op[j]);
                sdOT[(ch * 4 + j) * TS + qq] = dp[j];
                sQT[(ch * 4 + j) * TS + qq] = qp[j];
            }
            *reinterpret_cast<uint2*>(sdO + qq * DH + ch * 4) = d2;
            part += __shfl_xor(part, 1, 64);
            part += __shfl_xor(part, 2, 64);
            part += __shfl_xor(part, 4, 64);
            if (ch == 0) sD[qq] = part;
        }
        __syncthreads();

        // ---- phase A ----
        bf16x8_t dof[4];                       // A operand of dPd: dO[query xb*16 + c16][8g ..]
#pragma unroll
        for (int xb = 0; xb < 4; ++xb) dof[xb] = *reinterpret_cast<const bf16x8_t*>(sdO + (xb * 16 + c16) * DH + g * 8);
#pragma unroll
        for (int kb = 0; kb < NBW; ++kb) {
            const int key = (wave * NBW + kb) * 16 + c16;
            const bf16x8_t vf = *reinterpret_cast<const bf16x8_t*>(sV + key * DH + g * 8);
            unsigned short pcu[4][4], dcu[4][4];
#pragma unroll
            for (int xb = 0; xb < 4; ++xb)
#pragma unroll
                for (int r = 0; r < 4; ++r) { pcu[xb][r] = pnx[xb][r]; dcu[xb][r] = dnx[xb][r]; }
            if (kb + 1 < NBW) fetch(q0, kb + 1);
            else if (t + 1 < n_tiles) fetch(q0 + QT, 0);
            unsigned ds_pk[4][2], pd_pk[4][2];
#pragma unroll
            for (int xb = 0; xb < 4; ++xb) {
                const f32x4_t dpd = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dof[xb], vf, f32x4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                const float4 d4 = *reinterpret_cast<const float4*>(sD + xb * 16 + 4 * g);   // D of queries xb*16 + 4g + r
                const float dd[4] = {d4.x, d4.y, d4.z, d4.w};
                float dsv[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pv = bf2f(pcu[xb][r]);
                    const float dp = (prob_drop == nullptr || dcu[xb][r] != 0) ? dpd[r] * dscale : 0.f;
                    dsv[r] = pv * (dp - dd[r]);
                }
                ds_pk[xb][0] = pack2bf(dsv[0], dsv[1]); ds_pk[xb][1] = pack2bf(dsv[2], dsv[3]);
                pd_pk[xb][0] = (unsigned)dcu[xb][0] | ((unsigned)dcu[xb][1] << 16);
                pd_pk[xb][1] = (unsigned)dcu[xb][2] | ((unsigned)dcu[xb][3] << 16);
                *reinterpret_cast<uint2*>(sdST + key * TS + xb * 16 + 4 * g) = make_uint2(ds_pk[xb][0], ds_pk[xb][1]);
            }
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                union { unsigned u[4]; bf16x8_t v8; } sb, pb;
                sb.u[0] = ds_pk[2 * c][0]; sb.u[1] = ds_pk[2 * c][1]; sb.u[2] = ds_pk[2 * c + 1][0]; sb.u[3] = ds_pk[2 * c + 1][1];
                pb.u[0] = pd_pk[2 * c][0]; pb.u[1] = pd_pk[2 * c][1]; pb.u[2] = pd_pk[2 * c + 1][0]; pb.u[3] = pd_pk[2 * c + 1][1];
#pragma unroll
                for (int eb = 0; eb < 2; ++eb) {
                    // A operands: Q^T, dO^T [feature eb*16 + c16][query slots of chunk c] (re-read per key block: cheaper than 32 VGPRs)
                    union { uint2 u[2]; bf16x8_t v8; } a, d;
                    a.u[0] = *reinterpret_cast<const uint2*>(sQT + (eb * 16 + c16) * TS + 32 * c + 4 * g);
                    a.u[1] = *reinterpret_cast<const uint2*>(sQT + (eb * 16 + c16) * TS + 32 * c + 16 + 4 * g);
                    d.u[0] = *reinterpret_cast<const uint2*>(sdOT + (eb * 16 + c16) * TS + 32 * c + 4 * g);
                    d.u[1] = *reinterpret_cast<const uint2*>(sdOT + (eb * 16 + c16) * TS + 32 * c + 16 + 4 * g);
                    accK[kb][eb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v8, sb.v8, accK[kb][eb], 0, 0, 0);
                    accV[kb][eb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(d.v8, pb.v8, accV[kb][eb], 0, 0, 0);
                }
            }
        }
        __syncthreads();

        // ---- phase B: dQ block (queries qb*16 .., features eb*16 ..) = sum over key chunks of 32 ----
        {
            const int qb = wave & 3, eb = wave >> 2;
            f32x4_t acc = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
            for (int c = 0; c < SKP / 32; ++c) {
                const int k_lo = 32 * c + 4 * g + (c16 >> 2), k_hi = k_lo + 16;
                const int col4 = (c16 & 3) * 4;
                union { struct { s16x4_t a, b; } hh; bf16x8_t v8; } kf, sf;
                kf.hh.a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(sK + k_lo * DH + eb * 16 + col4));
                kf.hh.b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(sK + k_hi * DH + eb * 16 + col4));
                sf.hh.a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(sdST + k_lo * TS + qb * 16 + col4));
                sf.hh.b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(sdST + k_hi * TS + qb * 16 + col4));
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf.v8, sf.v8, acc, 0, 0, 0);
            }
            const int qi = q0 + qb * 16 + c16;
            if (qi < Sq)
                *reinterpret_cast<uint2*>(dq + ((size_t)b * Sq + qi) * lddq + h * DH + eb * 16 + g * 4) =
                    make_uint2(pack2bf(acc[0] * scale, acc[1] * scale), pack2bf(acc[2] * scale, acc[3] * scale));
        }
        __syncthreads();
    }
#pragma unroll
    for (int kb = 0; kb < NBW; ++kb) {
        const int key = (wave * NBW + kb) * 16 + c16;
        if (key < Sk) {
#pragma unroll
            for (int eb = 0; eb < 2; ++eb) {
                *reinterpret_cast<uint2*>(dk + ((size_t)b * Sk + key) * lddk + h * DH + eb * 16 + g * 4) =
                    make_uint2(pack2bf(accK[kb][eb][0] * scale, accK[kb][eb][1] * scale), pack2bf(accK[kb][eb][2] * scale, accK[kb][eb][3] * scale));
                *reinterpret_cast<uint2*>(dv + ((size_t)b * Sk + key) * lddv + h * DH + eb * 16 + g * 4) =
                    make_uint2(pack2bf(accV[kb][eb][0], accV[kb][eb][1]), pack2bf(accV[kb][eb][2], accV[kb][eb][3]));
            }
        }
    }
}

// Second backward variant for long query ranges (encoder self-attention, 416 x 416).  attn_bwd_kernel needs the probabilities as
// 2-byte gathers (lane = key, four query rows), which makes it texture-path bound when the whole score matrix is large.  Here the
// score-shaped work is done in the FORWARD layout (lane = query, four consecutive keys: 8-byte loads of P / Pd, D is a per-lane
// scalar, dQ accumulates from registers exactly like P V in the forward kernel), dS and Pd go to LDS query-major, and the key-side
// products dK += dS^T Q, dV += Pd^T dO read both operands k-major (ds_read_b64_tr_b16) in a second phase.
//   tile = 32 queries; phase A: wave = (16-query half, quarter of the key blocks); phase B: wave = key blocks w, w + 8, ...
template <int NB>   // 16-key blocks, multiple of 8: keys <= 16 * NB
__global__ __launch_bounds__(512) void attn_bwd_rows_kernel(const bf16_t* __restrict__ q, int ldq, const bf16_t* __restrict__ kmat, int ldk,
                                                            const bf16_t* __restrict__ v, int ldv, const bf16_t* __restrict__ prob,
                                                            const bf16_t* __restrict__ prob_drop, const bf16_t* __restrict__ ctx, int ldo,
                                                            const bf16_t* __restrict__ dctx, int lddo, int H, int Sq, int Sk, int ld,
                                                            float scale, float drop_p, bf16_t* __restrict__ dq, int lddq,
                                                            bf16_t* __restrict__ dk, int lddk, bf16_t* __restrict__ dv, int lddv,
                                                            float* __restrict__ part, const float* __restrict__ lse,
                                                            const unsigned char* __restrict__ key_pad, unsigned long long seed,
                                                            const unsigned long long* __restrict__ seed_dev) {
    // prob == nullptr: flash-style recomputation -- P = exp(scale q.k - max) * rsum per (query, key), the keep mask from the forward
    // kernel's hash (same seed, same element index row * ld + key); `dropping` then comes from drop_p alone
    constexpr int SKP = NB * 16, DH = 32, QT = 32, LS = SKP + 8, NBQ = NB / 4, NBW = NB / 8;
    extern __shared__ __attribute__((aligned(16))) bf16_t smem[];
    bf16_t* sV = smem;                        // [SKP][32]
    bf16_t* sK = sV + SKP * DH;               // [SKP][32]
    bf16_t* sdS = sK + SKP * DH;              // [QT][LS]
    bf16_t* sPd = sdS + QT * LS;              // [QT][LS]
    bf16_t* sdO = sPd + QT * LS;              // [QT][32]
    bf16_t* sQ = sdO + QT * DH;               // [QT][32]
    float* sdQ = reinterpret_cast<float*>(sQ + QT * DH);   // [4][QT][32] partial dQ of the four key quarters
    float* sD = sdQ + 4 * QT * DH;            // [QT]
    float* sL = sD + QT;                      // [QT][2] (row maximum, 1 / row sum) of the tile's queries (recompute mode)
    unsigned char* sDead = reinterpret_cast<unsigned char*>(sL + 2 * QT);   // [SKP] 1 = key masked (recompute mode)
    typedef __attribute__((address_space(3))) s16x4_t* lds_v4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c16 = lane & 15;
    const int bh = blockIdx.x, b = bh / H, h = bh - b * H;
    const bool recompute = prob == nullptr;
    const bool dropping = recompute ? drop_p > 0.f : prob_drop != nullptr;
    const float dscale = dropping ? 1.f / (1.f - drop_p) : 1.f;
    const unsigned thresh = dropping ? (unsigned)(drop_p * 4294967296.0) : 0u;
    if (seed_dev) seed += *seed_dev;
    const int qh = wave & 1, kq = wave >> 1;
    if (recompute)
        for (int kk = tid; kk < SKP; kk += 512) sDead[kk] = (kk >= Sk || (key_pad != nullptr && key_pad[(size_t)b * Sk + kk])) ? 1 : 0;

    for (int c = tid; c < SKP * 4; c += 512) {
        const int key = c >> 2, ch = c & 3;
        uint4 kv = make_uint4(0, 0, 0, 0), vv = make_uint4(0, 0, 0, 0);
        if (key < Sk) {
            kv = *reinterpret_cast<const uint4*>(kmat + ((size_t)b * Sk + key) * ldk + h * DH + ch * 8);
            vv = *reinterpret_cast<const uint4*>(v + ((size_t)b * Sk + key) * ldv + h * DH + ch * 8);
        }
        *reinterpret_cast<uint4*>(sK + key * DH + (ch << 3)) = kv;
        *reinterpret_cast<uint4*>(sV + key * DH + (ch << 3)) = vv;
    }
    f32x4_t accK[NBW][2], accV[NBW][2];
#pragma unroll
    for (int i = 0; i < NBW; ++i)
#pragma unroll
        for (int eb = 0; eb < 2; ++eb) { accK[i][eb] = f32x4_t{0.f, 0.f, 0.f, 0.f}; accV[i][eb] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }

    // probabilities of this lane's query for its key quarter, fetched one tile ahead
    uint2 pnx[NBQ], dnx[NBQ];
#pragma unroll
    for (int jj = 0; jj < NBQ; ++jj) pnx[jj] = dnx[jj] = make_uint2(0, 0);
    auto fetch = [&](int q0_) {
        if (recompute) return;
        const int qi = q0_ + qh * 16 + c16;
#pragma unroll
        for (int jj = 0; jj < NBQ; ++jj) {
            const int k0 = (kq * NBQ + jj) * 16 + 4 * g;
            uint2 pv = make_uint2(0, 0), pd = make_uint2(0, 0);
            if (qi < Sq && k0 < ld) {
                const size_t at = ((size_t)bh * Sq + qi) * ld + k0;
                pv = *reinterpret_cast<const uint2*>(prob + at);
                pd = prob_drop ? *reinterpret_cast<const uint2*>(prob_drop + at) : pv;
            }
            pnx[jj] = pv;
            dnx[jj] = pd;
        }
    };
    // gridDim.y workgroups share a head: each walks a contiguous run of query tiles (dQ rows are disjoint; their dK / dV sums go
    // to `part` [split][2][B*Sk][H*32] f32 and attn_bwd_fold_kernel adds them up)
    const int n_tiles_all = (Sq + QT - 1) / QT;
    const int per_split = (n_tiles_all + (int)gridDim.y - 1) / (int)gridDim.y;
    const int t_beg = (int)blockIdx.y * per_split;
    const int t_end = (t_beg + per_split < n_tiles_all) ? t_beg + per_split : n_tiles_all;
    fetch(t_beg * QT);
    // rows of dO, O, Q of the next tile (threads 0..255: query tid / 8, features 4 (tid % 8) ..), also fetched one tile ahead
    uint2 d2n = make_uint2(0, 0), o2n = make_uint2(0, 0), q2n = make_uint2(0, 0);
    float2 lsen = make_float2(0.f, 0.f);
    auto fetch_rows = [&](int q0_) {
        const int qi = q0_ + (tid >> 3), ch = tid & 7;
        d2n = o2n = q2n = make_uint2(0, 0);
        lsen = make_float2(0.f, 0.f);
        if (recompute && tid < 256 && ch == 0 && qi < Sq) lsen = *reinterpret_cast<const float2*>(lse + 2 * ((size_t)bh * Sq + qi));
        if (tid < 256 && qi < Sq) {
            d2n = *reinterpret_cast<const uint2*>(dctx + ((size_t)b * Sq + qi) * lddo + h * DH + ch * 4);
            o2n = *reinterpret_cast<const uint2*>(ctx + ((size_t)b * Sq + qi) * ldo + h * DH + ch * 4);
            q2n = *reinterpret_cast<const uint2*>(q + ((size_t)b * Sq + qi) * ldq + h * DH + ch * 4);
        }
    };
    fetch_rows(t_beg * QT);

    const int n_tiles = t_end;
    for (int t = t_beg; t < n_tiles; ++t) {
        const int q0 = t * QT;
        if (tid < 256) {   // ---- tile prologue: dO, Q rows and D = rowsum(dO o O) ----
            const int qq = tid >> 3, ch = tid & 7;
            const uint2 d2 = d2n, o2 = o2n, q2 = q2n;
            const bf16_t* dp = reinterpret_cast<const bf16_t*>(&d2);
            const bf16_t* op = reinterpret_cast<const bf16_t*>(&o2);
            float part = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) part += bf2f(dp[j]) * bf2f(op[j]);
            *reinterpret_cast<uint2*>(sdO + qq * DH + ch * 4) = d2;
            *reinterpret_cast<uint2*>(sQ + qq * DH + ch * 4) = q2;
            part += __shfl_xor(part, 1, 64);
            part += __shfl_xor(part, 2, 64);
            part += __shfl_xor(part, 4, 64);
            if (ch == 0) { sD[qq] = part; sL[2 * qq] = lsen.x; sL[2 * qq + 1] = lsen.y; }
        }
        __syncthreads();

        {   // ---- phase A: this wave's 16 queries x its quarter of the keys ----
            const int ql = qh * 16 + c16;                              // query of this lane inside the tile
            const bf16x8_t dof = *reinterpret_cast<const bf16x8_t*>(sdO + ql * DH + g * 8);
            const float dsum = sD[ql];
            const float mx_q = sL[2 * ql], rsum_q = sL[2 * ql + 1];
            const bool qlive = q0 + ql < Sq;
            const size_t prow = (size_t)bh * Sq + (qlive ? q0 + ql : 0);
            bf16x8_t qf = {0, 0, 0, 0, 0, 0, 0, 0};
            if (recompute) qf = *reinterpret_cast<const bf16x8_t*>(sQ + ql * DH + g * 8);
            uint2 pcu[NBQ], dcu[NBQ];
#pragma unroll
            for (int jj = 0; jj < NBQ; ++jj) { pcu[jj] = pnx[jj]; dcu[jj] = dnx[jj]; }
            if (t + 1 < n_tiles) { fetch(q0 + QT); fetch_rows(q0 + QT); }
            unsigned ds_pk[NBQ][2];
#pragma unroll
            for (int jj = 0; jj < NBQ; ++jj) {
                const int j = kq * NBQ + jj;
                const bf16x8_t vf = *reinterpret_cast<const bf16x8_t*>(sV + (j * 16 + c16) * DH + g * 8);
                const f32x4_t dpd = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, dof, f32x4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                float dsv[4];
                if (recompute) {
                    const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(sK + (j * 16 + c16) * DH + g * 8);
                    const f32x4_t sc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf, f32x4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                    const unsigned dead4 = *reinterpret_cast<const unsigned*>(sDead + j * 16 + g * 4);
                    const int k0 = j * 16 + 4 * g;
                    float pdv[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const bool dead = ((dead4 >> (8 * r)) & 0xffu) != 0 || !qlive;
                        const float pv = dead ? 0.f : __expf(scaled_minus(sc[r], scale, mx_q)) * rsum_q;
                        const bool keep = !dropping || dropout_keep(seed, (unsigned long long)prow * ld + k0 + r, thresh);
                        pdv[r] = keep ? pv * dscale : 0.f;
                        const float dp = keep ? dpd[r] * dscale : 0.f;
                        dsv[r] = pv * (dp - dsum);
                    }
                    dcu[jj] = make_uint2(pack2bf(pdv[0], pdv[1]), pack2bf(pdv[2], pdv[3]));
                } else {
                    const unsigned pw[2] = {pcu[jj].x, pcu[jj].y}, dw[2] = {dcu[jj].x, dcu[jj].y};
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const unsigned short pb = (unsigned short)(pw[r >> 1] >> (16 * (r & 1))), db = (unsigned short)(dw[r >> 1] >> (16 * (r & 1)));
                        const float dp = (!dropping || db != 0) ? dpd[r] * dscale : 0.f;
                        dsv[r] = bf2f(pb) * (dp - dsum);
                    }
                }
                ds_pk[jj][0] = pack2bf(dsv[0], dsv[1]);
                ds_pk[jj][1] = pack2bf(dsv[2], dsv[3]);
                *reinterpret_cast<uint2*>(sdS + ql * LS + j * 16 + 4 * g) = make_uint2(ds_pk[jj][0], ds_pk[jj][1]);
                *reinterpret_cast<uint2*>(sPd + ql * LS + j * 16 + 4 * g) = dcu[jj];
            }
            // dQ partial over this key quarter: slot 8g + r <-> key 32c + 4g + r, slot 8g + 4 + r <-> key 32c + 16 + 4g + r
            f32x4_t acc[2] = {f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int cc = 0; cc < NBQ / 2; ++cc) {
                union { unsigned u[4]; bf16x8_t v8; } sb;
                sb.u[0] = ds_pk[2 * cc][0]; sb.u[1] = ds_pk[2 * cc][1]; sb.u[2] = ds_pk[2 * cc + 1][0]; sb.u[3] = ds_pk[2 * cc + 1][1];
                const int c = (kq * NBQ) / 2 + cc;
                const int k_lo = 32 * c + 4 * g + (c16 >> 2), k_hi = k_lo + 16;
#pragma unroll
                for (int eb = 0; eb < 2; ++eb) {
                    union { struct { s16x4_t a, b; } hh; bf16x8_t v8; } kf;
                    kf.hh.a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(sK + k_lo * DH + eb * 16 + (c16 & 3) * 4));
                    kf.hh.b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(sK + k_hi * DH + eb * 16 + (c16 & 3) * 4));
                    acc[eb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf.v8, sb.v8, acc[eb], 0, 0, 0);
                }
            }
#pragma unroll
            for (int eb = 0; eb < 2; ++eb)
                *reinterpret_cast<float4*>(sdQ + ((size_t)kq * QT + ql) * DH + eb * 16 + 4 * g) = make_float4(acc[eb][0], acc[eb][1], acc[eb][2], acc[eb][3]);
        }
        __syncthreads();

        // ---- phase B ----
        if (tid < 256) {   // dQ of the tile = sum of the four key-quarter partials
            const int qq = tid >> 3, e4 = (tid & 7) * 4, qi = q0 + qq;
            float4 a = *reinterpret_cast<const float4*>(sdQ + ((size_t)0 * QT + qq) * DH + e4);
#pragma unroll
            for (int kk = 1; kk < 4; ++kk) {
                const float4 o = *reinterpret_cast<const float4*>(sdQ + ((size_t)kk * QT + qq) * DH + e4);
                a.x += o.x; a.y += o.y; a.z += o.z; a.w += o.w;
            }
            if (qi < Sq)
                *reinterpret_cast<uint2*>(dq + ((size_t)b * Sq + qi) * lddq + h * DH + e4) =
                    make_uint2(pack2bf(a.x * scale, a.y * scale), pack2bf(a.z * scale, a.w * scale));
        }
        {   // dK += dS^T Q, dV += Pd^T dO for this wave's key blocks (reduction over the tile's 32 queries)
            const int q_lo = 4 * g + (c16 >> 2), q_hi = q_lo + 16, col4 = (c16 & 3) * 4;
            union { struct { s16x4_t a, b; } hh; bf16x8_t v8; } qf[2], of[2];
#pragma unroll
            for (int eb = 0; eb < 2; ++eb) {
                qf[eb].hh.a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(sQ + q_lo * DH + eb * 16 + col4));
                qf[eb].hh.b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(sQ + q_hi * DH + eb * 16 + col4));
                of[eb].hh.a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(sdO + q_lo * DH + eb * 16 + col4));
                of[eb].hh.b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(sdO + q_hi * DH + eb * 16 + col4));
            }
#pragma unroll
            for (int i = 0; i < NBW; ++i) {
                const int jb = wave + 8 * i;
                union { struct { s16x4_t a, b; } hh; bf16x8_t v8; } sf, pf;
                sf.hh.a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(sdS + q_lo * LS + jb * 16 + col4));
                sf.hh.b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(sdS + q_hi * LS + jb * 16 + col4));
                pf.hh.a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(sPd + q_lo * LS + jb * 16 + col4));
                pf.hh.b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(sPd + q_hi * LS + jb * 16 + col4));
#pragma unroll
                for (int eb = 0; eb < 2; ++eb) {
                    accK[i][eb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[eb].v8, sf.v8, accK[i][eb], 0, 0, 0);
                    accV[i][eb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(of[eb].v8, pf.v8, accV[i][eb], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < NBW; ++i) {
        const int key = (wave + 8 * i) * 16 + c16;
        if (key < Sk) {
#pragma unroll
            for (int eb = 0; eb < 2; ++eb) {
                if (part != nullptr) {
                    const size_t plane = (size_t)(gridDim.x / H) * Sk * H * DH;            // B*Sk x H*32
                    // partial sums travel as bf16 (round 3): the four query-split partials of an encoder layer were 27.9 MB of fp32 written and
                    // read back per launch against 5 MB of dQ / dK / dV (profiles/r02_pmc_traffic.json); the fold adds them in fp32
                    bf16_t* const pb = reinterpret_cast<bf16_t*>(part);
                    bf16_t* pk = pb + ((size_t)blockIdx.y * 2 + 0) * plane + ((size_t)b * Sk + key) * (H * DH) + h * DH + eb * 16 + g * 4;
                    bf16_t* pv = pb + ((size_t)blockIdx.y * 2 + 1) * plane + ((size_t)b * Sk + key) * (H * DH) + h * DH + eb * 16 + g * 4;
                    *reinterpret_cast<uint2*>(pk) = make_uint2(pack2bf(accK[i][eb][0], accK[i][eb][1]), pack2bf(accK[i][eb][2], accK[i][eb][3]));
                    *reinterpret_cast<uint2*>(pv) = make_uint2(pack2bf(accV[i][eb][0], accV[i][eb][1]), pack2bf(accV[i][eb][2], accV[i][eb][3]));
                } else {
                    *reinterpret_cast<uint2*>(dk + ((size_t)b * Sk + key) * lddk + h * DH + eb * 16 + g * 4) =
                        make_uint2(pack2bf(accK[i][eb][0] * scale, accK[i][eb][1] * scale), pack2bf(accK[i][eb][2] * scale, accK[i][eb][3] * scale));
                    *reinterpret_cast<uint2*>(dv + ((size_t)b * Sk + key) * lddv + h * DH + eb * 16 + g * 4) =
                        make_uint2(pack2bf(accV[i][eb][0], accV[i][eb][1]), pack2bf(accV[i][eb][2], accV[i][eb][3]));
                }
            }
        }
    }
}

// dk = scale * sum_s part[s][0], dv = sum_s part[s][1]: rows = B*Sk, d = H*32 columns, 4 columns per thread
__global__ __launch_bounds__(256) void attn_bwd_fold_kernel(const float* __restrict__ part_, int splits, long long rows, int d, float scale,
                                                            bf16_t* __restrict__ dk, int lddk, bf16_t* __restrict__ dv, int lddv) {
    const bf16_t* const part = reinterpret_cast<const bf16_t*>(part_);     // bf16 partials, 8 columns (16 bytes) per thread
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const int c8 = d >> 3;
    if (idx >= rows * c8) return;
    const long long r = idx / c8;
    const int c = (int)(idx - r * c8) * 8;
    const size_t plane = (size_t)rows * d;
    float ak[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, av[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int sp = 0; sp < splits; ++sp) {
        const uint4 a = *reinterpret_cast<const uint4*>(part + ((size_t)sp * 2 + 0) * plane + (size_t)r * d + c);
        const uint4 b = *reinterpret_cast<const uint4*>(part + ((size_t)sp * 2 + 1) * plane + (size_t)r * d + c);
        const unsigned aw[4] = {a.x, a.y, a.z, a.w}, bw[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            ak[2 * q] += __uint_as_float(aw[q] << 16); ak[2 * q + 1] += __uint_as_float(aw[q] & 0xffff0000u);
            av[2 * q] += __uint_as_float(bw[q] << 16); av[2 * q + 1] += __uint_as_float(bw[q] & 0xffff0000u);
        }
    }
    *reinterpret_cast<uint4*>(dk + (size_t)r * lddk + c) = make_uint4(pack2bf(ak[0] * scale, ak[1] * scale), pack2bf(ak[2] * scale, ak[3] * scale),
                                                                    pack2bf(ak[4] * scale, ak[5] * scale), pack2bf(ak[6] * scale, ak[7] * scale));
    *reinterpret_cast<uint4*>(dv + (size_t)r * lddv + c) = make_uint4(pack2bf(av[0], av[1]), pack2bf(av[2], av[3]), pack2bf(av[4], av[5]), pack2bf(av[6], av[7]));
}

}  // namespace toist

using namespace toist;

extern "C" int toist_attn_fwd(const void* q, int ldq, const void* kmat, int ldk, const void* v, int ldv, const uint8_t* key_pad, int B, int H, int Sq,
                              int Sk, int dh, int ld, float scale, void* prob, void* prob_drop, float drop_p, uint64_t seed, const uint64_t* seed_dev,
                              void* ctx, int ldo, float* lse, void* stream) {
    TOIST_REQUIRE(q && kmat && v && (prob || lse) && ctx && B > 0 && H > 0 && Sq > 0 && Sk > 0, "toist_attn_fwd: bad args (prob or lse must be given)");
    TOIST_REQUIRE(prob != nullptr || prob_drop == nullptr, "toist_attn_fwd: prob_drop without prob");
    TOIST_REQUIRE(prob == nullptr || drop_p == 0.f || prob_drop != nullptr, "toist_attn_fwd: with prob and dropout the dropped-out copy prob_drop is required");
    TOIST_REQUIRE(dh == 32, "toist_attn_fwd: head dim must be 32 (got %d)", dh);
    TOIST_REQUIRE(Sk <= 480 && ld >= Sk && (ld % 8) == 0, "toist_attn_fwd: Sk <= 480 (K, V and the mask of a head must fit 64 KB of LDS) and ld = round8(Sk) (got %d, %d)", Sk, ld);
    TOIST_REQUIRE((ldq % 8) == 0 && (ldk % 8) == 0 && (ldv % 8) == 0 && (ldo % 4) == 0, "toist_attn_fwd: row strides must keep 16-byte alignment");
    TOIST_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "toist_attn_fwd: bad dropout p");
    const dim3 grid((Sq + 63) / 64, B * H), block(256);
    hipStream_t st = (hipStream_t)stream;
#define TOIST_ATTN(NB)                                                                                                                       \
    hipLaunchKernelGGL((attn_fwd_kernel<NB>), grid, block, 2 * (NB) * 16 * 32 * sizeof(bf16_t) + (NB) * 16, st, (const bf16_t*)q, ldq, (const bf16_t*)kmat, ldk, \
                       (const bf16_t*)v, ldv, key_pad, H, Sq, Sk, ld, scale, (bf16_t*)prob, (bf16_t*)prob_drop, drop_p,                      \
                       (unsigned long long)seed, (const unsigned long long*)seed_dev, (bf16_t*)ctx, ldo, lse)
    if (Sk <= 128) TOIST_ATTN(8);
    else if (Sk <= 416) TOIST_ATTN(26);
    else TOIST_ATTN(30);
#undef TOIST_ATTN
    return check_launch("toist_attn_fwd");
}

extern "C" int toist_attn_bwd(const void* q, int ldq, const void* kmat, int ldk, const void* v, int ldv, const void* prob, const void* prob_drop,
                              const void* ctx, int ldo, const void* dctx, int lddo, int B, int H, int Sq, int Sk, int dh, int ld, float scale,
                              float drop_p, void* dq, int lddq, void* dk, int lddk, void* dv, int lddv, int variant, float* workspace,
                              int q_splits, const float* lse, const uint8_t* key_pad, uint64_t seed, const uint64_t* seed_dev, void* stream) {
    TOIST_REQUIRE(q && kmat && v && (prob || lse) && ctx && dctx && dq && dk && dv && B > 0 && H > 0 && Sq > 0 && Sk > 0, "toist_attn_bwd: bad args");
    TOIST_REQUIRE(prob != nullptr || variant != 1, "toist_attn_bwd: the recomputing (lse) mode exists for the query-major variant only");
    if (prob == nullptr) variant = 2;
    TOIST_REQUIRE(variant >= 0 && variant <= 2, "toist_attn_bwd: variant 0 (auto), 1 (key-major) or 2 (query-major)");
    TOIST_REQUIRE((ld % 8) == 0, "toist_attn_bwd: ld must be a multiple of 8");
    TOIST_REQUIRE(q_splits >= 1 && q_splits <= 16 && (q_splits == 1 || workspace != nullptr), "toist_attn_bwd: 1..16 query splits; > 1 needs a workspace");
    TOIST_REQUIRE(dh == 32, "toist_attn_bwd: head dim must be 32 (got %d)", dh);
    TOIST_REQUIRE(Sk <= 512 && ld >= Sk, "toist_attn_bwd: Sk <= 512 and ld >= Sk (got %d, %d)", Sk, ld);
    TOIST_REQUIRE((ldq % 4) == 0 && (ldk % 8) == 0 && (ldv % 8) == 0 && (ldo % 4) == 0 && (lddo % 4) == 0 && (lddq % 4) == 0 && (lddk % 4) == 0 &&
                      (lddv % 4) == 0, "toist_attn_bwd: row strides must keep 8-byte (q, ctx, gradients) / 16-byte (k, v) alignment");
    TOIST_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "toist_attn_bwd: bad dropout p");
    hipStream_t st = (hipStream_t)stream;
#define TOIST_ATTN_BWD(NBW)                                                                                                                  \
    do {                                                                                                                                     \
        const size_t lds = sizeof(bf16_t) * ((size_t)2 * (8 * NBW * 16) * 32 + (size_t)(8 * NBW * 16) * 72 + 64 * 32 + 2 * 32 * 72) + 64 * 4;  \
        if (lds > 64 * 1024) {                                                                                                               \
            hipError_t e = hipFuncSetAttribute((const void*)attn_bwd_kernel<NBW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);     \
            if (e != hipSuccess) { set_last_error("toist_attn_bwd: set LDS size: %s", hipGetErrorString(e)); return TOIST_EHIP; }            \
        }                                                                                                                                    \
        hipLaunchKernelGGL((attn_bwd_kernel<NBW>), dim3(B * H), dim3(512), lds, st, (const bf16_t*)q, ldq, (const bf16_t*)kmat, ldk,          \
                           (const bf16_t*)v, ldv, (const bf16_t*)prob, (const bf16_t*)prob_drop, (const bf16_t*)ctx, ldo, (const bf16_t*)dctx,  \
                           lddo, H, Sq, Sk, ld, scale, drop_p, (bf16_t*)dq, lddq, (bf16_t*)dk, lddk, (bf16_t*)dv, lddv);                       \
    } while (0)
#define TOIST_ATTN_BWD_ROWS(NB)                                                                                                              \
    do {                                                                                                                                     \
        const size_t lds = sizeof(bf16_t) * ((size_t)2 * (NB * 16) * 32 + (size_t)2 * 32 * (NB * 16 + 8) + 2 * 32 * 32) + 4 * (4 * 32 * 32 + 32 + 64) + (NB) * 16; \
        if (lds > 64 * 1024) {                                                                                                               \
            hipError_t e = hipFuncSetAttribute((const void*)attn_bwd_rows_kernel<NB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            if (e != hipSuccess) { set_last_error("toist_attn_bwd: set LDS size: %s", hipGetErrorString(e)); return TOIST_EHIP; }            \
        }                                                                                                                                    \
        hipLaunchKernelGGL((attn_bwd_rows_kernel<NB>), dim3(B * H, q_splits), dim3(512), lds, st, (const bf16_t*)q, ldq, (const bf16_t*)kmat, \
                           ldk, (const bf16_t*)v, ldv, (const bf16_t*)prob, (const bf16_t*)prob_drop, (const bf16_t*)ctx, ldo,                \
                           (const bf16_t*)dctx, lddo, H, Sq, Sk, ld, scale, drop_p, (bf16_t*)dq, lddq, (bf16_t*)dk, lddk, (bf16_t*)dv, lddv,  \
                           q_splits > 1 ? workspace : (float*)nullptr, lse, key_pad, (unsigned long long)seed,                              \
                           (const unsigned long long*)seed_dev);                                                                           \
    } while (0)
    if (variant == 0) variant = (Sq > 128 && Sk > 128) ? 2 : 1;
    if (variant == 2) {
        if (q_splits > 1)       // checked BEFORE the rows kernel runs: an invalid call does no work
            TOIST_REQUIRE((lddk % 8) == 0 && (lddv % 8) == 0 && ((((size_t)dk) | ((size_t)dv)) & 15) == 0,
                          "toist_attn_bwd: query splits fold into 16-byte chunks of dk / dv (row strides %% 8, 16-byte aligned bases)");
        if (Sk <= 128) TOIST_ATTN_BWD_ROWS(8);
        else if (Sk <= 256) TOIST_ATTN_BWD_ROWS(16);
        else TOIST_ATTN_BWD_ROWS(32);
        if (q_splits > 1) {
            const int rc2 = check_launch("toist_attn_bwd");
            if (rc2 != TOIST_OK) return rc2;
            const long long rows = (long long)B * Sk, n8 = rows * (H * 32 / 8);
            hipLaunchKernelGGL(attn_bwd_fold_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, st, workspace, q_splits, rows, H * 32, scale,
                               (bf16_t*)dk, lddk, (bf16_t*)dv, lddv);
        }
    } else {
        TOIST_REQUIRE(q_splits == 1, "toist_attn_bwd: query splits need the query-major variant");
        if (Sk <= 128) TOIST_ATTN_BWD(1);
        else if (Sk <= 256) TOIST_ATTN_BWD(2);
        else TOIST_ATTN_BWD(4);
    }
#undef TOIST_ATTN_BWD
#undef TOIST_ATTN_BWD_ROWS
    return check_launch("toist_attn_bwd");
}
