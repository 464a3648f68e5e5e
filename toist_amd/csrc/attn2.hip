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
    const unsigned a = pair ^ s0;
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
constexpr int A2_TS = 24;      // row stride (bf16) of a wave's [32 keys][16 queries] dS tile
constexpr int A2_KS = 40;      // row stride (bf16) of a wave's [32 keys][32 features] K tile (start-up transpose)

constexpr int A2_NG = 2;       // query groups per workgroup: tiles are dealt round-robin to A2_NG sets of four waves (two waves per SIMD)

__global__ __launch_bounds__(256 * A2_NG) void attn2_bwd_kernel(const bf16_t* __restrict__ q, int ldq, const bf16_t* __restrict__ kmat, int ldk,
                                                        const bf16_t* __restrict__ v, int ldv, const bf16_t* __restrict__ ctx, int ldo,
                                                        const bf16_t* __restrict__ dctx, int lddo, const float* __restrict__ lse,
                                                        const unsigned char* __restrict__ key_pad, int H, int Sq, int Sk, int ldp, float scale,
                                                        float drop_p, unsigned long long seed, const unsigned long long* __restrict__ seed_dev,
                                                        bf16_t* __restrict__ dq, int lddq, bf16_t* __restrict__ dk, int lddk,
                                                        bf16_t* __restrict__ dv, int lddv, bf16_t* __restrict__ dq_part, long long part_stride) {
    // A lone wave on its SIMD issues about one instruction per five cycles, and a query tile is ~1400 dependent instructions: with four
    // waves per workgroup and one workgroup per CU the first version of this kernel was bound by exactly that (31.8 us for the encoder
    // shape, profiles/r04_attn_core_us.txt).  The workgroup therefore carries A2_NG groups of four waves; every group owns the SAME keys
    // (wave w of each group: pair 4 x + w) and every A2_NG-th query tile, with its own staging buffers and dQ slabs; the groups' dK / dV
    // sums meet in LDS once, at the end.
    constexpr int DH = 32, QT = 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char a2_smem[];
    bf16_t* const sQ_ = reinterpret_cast<bf16_t*>(a2_smem);                                  // [NG][2][QT * DH]
    bf16_t* const sdO_ = sQ_ + A2_NG * 2 * QT * DH;                                          // [NG][2][QT * DH]
    bf16_t* const sT_ = sdO_ + A2_NG * 2 * QT * DH;                                          // [NG * 4][2 * 32 * A2_TS] wave-private
    float* const sD_ = reinterpret_cast<float*>(sT_ + A2_NG * 4 * 2 * 32 * A2_TS);           // [NG][2][QT]
    float* const sL_ = sD_ + A2_NG * 2 * QT;                                                 // [NG][2][QT * 2]
    float* const sSlab_ = sL_ + A2_NG * 2 * QT * 2;                                          // [NG][2][4][QT * DH]
    if (seed_dev) seed += *seed_dev;
    const unsigned s0 = (unsigned)seed, s1 = (unsigned)(seed >> 32) ^ ((unsigned)seed * 0x9E3779B9u);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c16 = lane & 15;
    const int grp = wave >> 2, w4 = wave & 3, gtid = tid & 255;
    const int bh = blockIdx.y, b = bh / H, h = bh - b * H;
    const int nkp = (Sk + 31) >> 5;                           // 32-key pairs of blocks in this head
    const int kp = blockIdx.x * 4 + w4;                       // this wave's pair
    const bool active = kp < nkp;
    const int nw = nkp - (int)blockIdx.x * 4 < 4 ? nkp - (int)blockIdx.x * 4 : 4;     // active waves of a group (>= 1)
    const float c = scale * LOG2E;
    const bool dropping = drop_p > 0.f;
    const unsigned t32 = dropping ? ((unsigned)(drop_p * 65536.0f + 0.5f)) << 16 : 0u;
    const float dscale = dropping ? 1.f / (1.f - drop_p) : 1.f;
    const int par = c16 & 1;                                  // key parity = the 16-bit field of a pair hash this lane reads
    const unsigned fsh = par ? 0u : 16u;                      // field -> bits 16-31: compare (h << fsh) with t << 16
    bf16_t* const sT = sT_ + wave * (2 * 32 * A2_TS);
    bf16_t* const sQg = sQ_ + grp * (2 * QT * DH);
    bf16_t* const sdOg = sdO_ + grp * (2 * QT * DH);
    float* const sDg = sD_ + grp * (2 * QT);
    float* const sLg = sL_ + grp * (2 * QT * 2);
    float* const sSlabg = sSlab_ + grp * (2 * 4 * QT * DH);

    // ---- this wave's keys: B fragments of K and V (lane = key c16 of block kb, features 8g ..), K^T as A fragments ----
    bf16x8_t kfB[2], vfB[2], ktA[2];
    bool dead[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        const int key = kp * 32 + kb * 16 + c16;
        const bool in = active && key < Sk;
        dead[kb] = !in || (key_pad != nullptr && key_pad[(size_t)b * Sk + key] != 0);
        kfB[kb] = bf16x8_t{0, 0, 0, 0, 0, 0, 0, 0};
        vfB[kb] = bf16x8_t{0, 0, 0, 0, 0, 0, 0, 0};
        if (in) {
            kfB[kb] = *reinterpret_cast<const bf16x8_t*>(kmat + ((size_t)b * Sk + key) * ldk + h * DH + g * 8);
            vfB[kb] = *reinterpret_cast<const bf16x8_t*>(v + ((size_t)b * Sk + key) * ldv + h * DH + g * 8);
        }
        *reinterpret_cast<bf16x8_t*>(sT + (kb * 16 + c16) * A2_KS + g * 8) = kfB[kb];       // [32 keys][32 features] for the transpose
    }
    const bool any_dead = __any(dead[0] || dead[1]);
#pragma unroll
    for (int eb = 0; eb < 2; ++eb) {          // lane (feature 16 eb + c16, group g): keys 8g .. 8g + 7 of the pair
        const bf16_t* base = sT + (8 * g + (c16 >> 2)) * A2_KS + eb * 16 + (c16 & 3) * 4;
        ktA[eb] = tr_pair(base, base + 4 * A2_KS);
    }
    f32x4_t accK[2][2], accV[2][2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int eb = 0; eb < 2; ++eb) { accK[kb][eb] = f32x4_t{0.f, 0.f, 0.f, 0.f}; accV[kb][eb] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }

    // ---- query tiles: rows of dO, O, Q (thread = query gtid / 8, features 4 (gtid % 8) ..) travel one tile (of this group) ahead ----
    const int n_tiles = (Sq + QT - 1) / QT;
    const int n_iter = (n_tiles + A2_NG - 1) / A2_NG;
    struct Rows { u32x2_t d2, o2, q2; float2 ls; };       // one thread's piece of a tile's dO / O / Q rows + (row maximum, 1 / row sum)
    Rows ra, rb;                                           // loaded TWO iterations ahead: a tile's loads have a whole iteration to land
    // row pointers of this thread's query in the group's first tile, advanced by A2_NG tiles per fetch: no multiplications inside the loop
    const int qrow0 = grp * QT + (gtid >> 3);
    const bf16_t* pd = dctx + ((size_t)b * Sq + qrow0) * lddo + h * DH + (gtid & 7) * 4;
    const bf16_t* po = ctx + ((size_t)b * Sq + qrow0) * ldo + h * DH + (gtid & 7) * 4;
    const bf16_t* pq = q + ((size_t)b * Sq + qrow0) * ldq + h * DH + (gtid & 7) * 4;
    const float* pl = lse + 2 * ((size_t)bh * Sq + qrow0);
    const size_t sd = (size_t)A2_NG * QT * lddo, so = (size_t)A2_NG * QT * ldo, sq_ = (size_t)A2_NG * QT * ldq;
    // where this thread's 4 features of query gtid / 8 of dQ go: the workgroup's share (key splits) or dq itself
    const int ldout = dq_part != nullptr ? H * DH : lddq;
    bf16_t* pout = (dq_part != nullptr ? dq_part + (size_t)blockIdx.x * part_stride : dq) + ((size_t)b * Sq + qrow0) * ldout + h * DH + (gtid & 7) * 4;
    const size_t sout = (size_t)A2_NG * QT * ldout;
    auto fetch_rows = [&](Rows& r, int q0_) {
        const int qi = q0_ + (gtid >> 3), ch = gtid & 7;
        r.d2 = u32x2_t{0, 0}; r.o2 = u32x2_t{0, 0}; r.q2 = u32x2_t{0, 0};
        r.ls = make_float2(0.f, 0.f);
        if (qi < Sq) {
            r.d2 = *reinterpret_cast<const u32x2_t*>(pd);
            r.o2 = *reinterpret_cast<const u32x2_t*>(po);
            r.q2 = *reinterpret_cast<const u32x2_t*>(pq);
            if (ch == 0) r.ls = *reinterpret_cast<const float2*>(pl);
        }
        pd += sd; po += so; pq += sq_; pl += 2 * A2_NG * QT;
    };
    auto stage_rows = [&](const Rows& r, int buf) {
        const int qq = gtid >> 3, ch = gtid & 7;
        float part = 0.f;
#pragma unroll
        for (int w2 = 0; w2 < 2; ++w2) {
            const unsigned dw = r.d2[w2], ow = r.o2[w2];
            part += __uint_as_float(dw << 16) * __uint_as_float(ow << 16) + __uint_as_float(dw & 0xffff0000u) * __uint_as_float(ow & 0xffff0000u);
        }
        *reinterpret_cast<u32x2_t*>(sdOg + buf * (QT * DH) + qq * DH + ch * 4) = r.d2;
        *reinterpret_cast<u32x2_t*>(sQg + buf * (QT * DH) + qq * DH + ch * 4) = r.q2;
        part += __shfl_xor(part, 1, 64);
        part += __shfl_xor(part, 2, 64);
        part += __shfl_xor(part, 4, 64);
        if (ch == 0) {
            sDg[buf * QT + qq] = part;
            sLg[buf * (QT * 2) + 2 * qq] = (r.ls.x == -INFINITY) ? 0.f : r.ls.x;
            sLg[buf * (QT * 2) + 2 * qq + 1] = r.ls.y;
        }
    };
    fetch_rows(ra, grp * QT);                               // iteration 0 (staged now)
    fetch_rows(rb, grp * QT + A2_NG * QT);                  // iteration 1 (staged at the end of iteration 0)
    stage_rows(ra, 0);
    __syncthreads();

    const unsigned hl = (unsigned)(ldp >> 1);                        // pairs per score row
    // pair index of this lane's element (query 4g + 2 par of tile 0, key of block 0): + (q0 + 16 blk) hl + 8 kb per (tile, query block, key block)
    const unsigned pair_lane = ((unsigned)bh * (unsigned)Sq + (unsigned)(4 * g + 2 * par)) * hl + (unsigned)((kp * 32 + c16) >> 1);
    auto iteration = [&](const int it, Rows& r_next, Rows& r_next2) {      // r_next: rows of iteration it + 1 (in flight since it - 1); r_next2 <- it + 2
        const int t = it * A2_NG + grp;
        const int buf = it & 1, q0 = t * QT;
        const bool valid = t < n_tiles;
        if (it + 2 < n_iter) fetch_rows(r_next2, q0 + 2 * A2_NG * QT);        // r_next2 was staged one iteration ago: free
        const bf16_t* const sQ = sQg + buf * (QT * DH);
        const bf16_t* const sdO = sdOg + buf * (QT * DH);
        float* const slab = sSlabg + (buf * 4 + w4) * (QT * DH);
        if (active && valid) {
            // ---- scores and dP with lane = key: rows (queries) 4g .. 4g + 3 of each 16-query block ----
            unsigned ds_pk[2][2][2], pd_pk[2][2][2];       // [query block][key block][pair of queries]
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                const bf16x8_t qa = *reinterpret_cast<const bf16x8_t*>(sQ + (blk * 16 + c16) * DH + g * 8);
                const bf16x8_t da = *reinterpret_cast<const bf16x8_t*>(sdO + (blk * 16 + c16) * DH + g * 8);
                const float4 D4 = *reinterpret_cast<const float4*>(sDg + buf * QT + blk * 16 + 4 * g);
                const float4 La = *reinterpret_cast<const float4*>(sLg + buf * (QT * 2) + 2 * (blk * 16 + 4 * g));
                const float4 Lb = *reinterpret_cast<const float4*>(sLg + buf * (QT * 2) + 2 * (blk * 16 + 4 * g) + 4);
                const float Dq[4] = {D4.x, D4.y, D4.z, D4.w}, mcq[4] = {La.x, La.z, Lb.x, Lb.z}, rsq[4] = {La.y, La.w, Lb.y, Lb.w};
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    f32x4_t sc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa, kfB[kb], f32x4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                    const f32x4_t dpd = __builtin_amdgcn_mfma_f32_16x16x32_bf16(da, vfB[kb], f32x4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                    if (any_dead && dead[kb]) sc = f32x4_t{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
                    unsigned hh[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
                    if (dropping) {
                        // the lanes of an (even, odd) key pair share their hashes: each computes two of the four queries, a DPP swap delivers the rest
                        const unsigned pair0 = pair_lane + (unsigned)(q0 + blk * 16) * hl + (unsigned)(kb * 8);
                        const unsigned m0 = pair_hash(pair0, s0, s1), m1 = pair_hash(pair0 + hl, s0, s1);
                        const unsigned o0 = (unsigned)__builtin_amdgcn_mov_dpp((int)m0, 0xB1, 0xf, 0xf, true);      // quad_perm [1, 0, 3, 2]
                        const unsigned o1 = (unsigned)__builtin_amdgcn_mov_dpp((int)m1, 0xB1, 0xf, 0xf, true);
                        hh[0] = (par ? o0 : m0) << fsh;
                        hh[1] = (par ? o1 : m1) << fsh;
                        hh[2] = (par ? m0 : o0) << fsh;
                        hh[3] = (par ? m1 : o1) << fsh;
                    }
                    float dsv[4], pdv[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float p = __builtin_amdgcn_exp2f((sc[r] - mcq[r]) * c) * rsq[r];
                        const bool keep = hh[r] >= t32;
                        pdv[r] = keep ? p * dscale : 0.f;
                        const float dp = keep ? dpd[r] * dscale : 0.f;
                        dsv[r] = p * (dp - Dq[r]);
                    }
                    ds_pk[blk][kb][0] = pack2bf(dsv[0], dsv[1]);
                    ds_pk[blk][kb][1] = pack2bf(dsv[2], dsv[3]);
                    pd_pk[blk][kb][0] = pack2bf(pdv[0], pdv[1]);
                    pd_pk[blk][kb][1] = pack2bf(pdv[2], pdv[3]);
                    *reinterpret_cast<u32x2_t*>(sT + blk * (32 * A2_TS) + (kb * 16 + c16) * A2_TS + 4 * g) = u32x2_t{ds_pk[blk][kb][0], ds_pk[blk][kb][1]};
                }
            }
            // ---- dK^T += Q^T dS, dV^T += dO^T Pd (reduction over the tile's 32 queries: slot 8g + r <-> query 4g + r, 8g + 4 + r <-> 16 + 4g + r) ----
            {
                const int q_lo = 4 * g + (c16 >> 2), col4 = (c16 & 3) * 4;
                bf16x8_t qT[2], oT[2];
#pragma unroll
                for (int eb = 0; eb < 2; ++eb) {
                    qT[eb] = tr_pair(sQ + q_lo * DH + eb * 16 + col4, sQ + (q_lo + 16) * DH + eb * 16 + col4);
                    oT[eb] = tr_pair(sdO + q_lo * DH + eb * 16 + col4, sdO + (q_lo + 16) * DH + eb * 16 + col4);
                }
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    const bf16x8_t sb = frag_of(ds_pk[0][kb][0], ds_pk[0][kb][1], ds_pk[1][kb][0], ds_pk[1][kb][1]);
                    const bf16x8_t pb = frag_of(pd_pk[0][kb][0], pd_pk[0][kb][1], pd_pk[1][kb][0], pd_pk[1][kb][1]);
#pragma unroll
                    for (int eb = 0; eb < 2; ++eb) {
                        accK[kb][eb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qT[eb], sb, accK[kb][eb], 0, 0, 0);
                        accV[kb][eb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(oT[eb], pb, accV[kb][eb], 0, 0, 0);
                    }
                }
            }
            // ---- this wave's share of dQ^T = K^T dS^T: dS read back transposed (lane = query c16, keys 8g .. 8g + 7 of the pair) ----
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                const bf16_t* base = sT + blk * (32 * A2_TS) + (8 * g + (c16 >> 2)) * A2_TS + (c16 & 3) * 4;
                const bf16x8_t sb = tr_pair(base, base + 4 * A2_TS);
#pragma unroll
                for (int eb = 0; eb < 2; ++eb) {
                    const f32x4_t r4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ktA[eb], sb, f32x4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                    *reinterpret_cast<float4*>(slab + (blk * 16 + c16) * DH + eb * 16 + 4 * g) = make_float4(r4[0], r4[1], r4[2], r4[3]);
                }
            }
        }
        if (it + 1 < n_iter) stage_rows(r_next, buf ^ 1);
        __syncthreads();     // the tiles' dQ shares are complete; the next tiles' rows are staged
        {   // dQ of this group's tile (this workgroup's keys): sum of the active waves' shares
            const int qq = gtid >> 3, e4 = (gtid & 7) * 4, qi = q0 + qq;
            const float* sl = sSlabg + buf * (4 * QT * DH) + qq * DH + e4;
            float4 a = *reinterpret_cast<const float4*>(sl);
            for (int w2 = 1; w2 < nw; ++w2) {
                const float4 o = *reinterpret_cast<const float4*>(sl + w2 * (QT * DH));
                a.x += o.x; a.y += o.y; a.z += o.z; a.w += o.w;
            }
            if (valid && qi < Sq) *reinterpret_cast<u32x2_t*>(pout) = u32x2_t{pack2bf(a.x * scale, a.y * scale), pack2bf(a.z * scale, a.w * scale)};
            pout += sout;
        }
    };
    for (int it = 0; it < n_iter; it += 2) {
        iteration(it, rb, ra);                       // even: stages rb (rows of it + 1), refills ra with the rows of it + 2
        if (it + 1 < n_iter) iteration(it + 1, ra, rb);
    }
    // ---- dK / dV: the groups' sums meet in LDS (the slabs are free after the last tile's reduction) ----
    __syncthreads();
    float* const xch = sSlab_ + w4 * (2 * 2 * 2 * 64 * 4);          // [w4][K | V][kb][eb][lane][4]
    if (grp > 0 && active) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int eb = 0; eb < 2; ++eb) {
                *reinterpret_cast<float4*>(xch + (((0 * 2 + kb) * 2 + eb) * 64 + lane) * 4) = make_float4(accK[kb][eb][0], accK[kb][eb][1], accK[kb][eb][2], accK[kb][eb][3]);
                *reinterpret_cast<float4*>(xch + (((1 * 2 + kb) * 2 + eb) * 64 + lane) * 4) = make_float4(accV[kb][eb][0], accV[kb][eb][1], accV[kb][eb][2], accV[kb][eb][3]);
            }
    }
    __syncthreads();
    if (grp == 0 && active) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const int key = kp * 32 + kb * 16 + c16;
#pragma unroll
            for (int eb = 0; eb < 2; ++eb) {
                const float4 xk = *reinterpret_cast<const float4*>(xch + (((0 * 2 + kb) * 2 + eb) * 64 + lane) * 4);
                const float4 xv = *reinterpret_cast<const float4*>(xch + (((1 * 2 + kb) * 2 + eb) * 64 + lane) * 4);
                const float k4[4] = {accK[kb][eb][0] + xk.x, accK[kb][eb][1] + xk.y, accK[kb][eb][2] + xk.z, accK[kb][eb][3] + xk.w};
                const float v4[4] = {accV[kb][eb][0] + xv.x, accV[kb][eb][1] + xv.y, accV[kb][eb][2] + xv.z, accV[kb][eb][3] + xv.w};
                if (key < Sk) {
                    *reinterpret_cast<u32x2_t*>(dk + ((size_t)b * Sk + key) * lddk + h * DH + eb * 16 + g * 4) =
                        u32x2_t{pack2bf(k4[0] * scale, k4[1] * scale), pack2bf(k4[2] * scale, k4[3] * scale)};
                    *reinterpret_cast<u32x2_t*>(dv + ((size_t)b * Sk + key) * lddv + h * DH + eb * 16 + g * 4) = u32x2_t{pack2bf(v4[0], v4[1]), pack2bf(v4[2], v4[3])};
                }
            }
        }
    }
}

constexpr size_t A2_BWD_LDS = (size_t)A2_NG * 2 * 32 * 32 * 2 * 2 + (size_t)A2_NG * 4 * 2 * 32 * A2_TS * 2 + (size_t)A2_NG * 2 * 32 * 4 + (size_t)A2_NG * 2 * 64 * 4 +
                              (size_t)A2_NG * 2 * 4 * 32 * 32 * 4;

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
