// Body of the key-owning attention backward (see csrc/attn2.hip for the algorithm): shared by attn2_bwd_kernel (one launch per attention block)
// and by the XCD-resident decoder backward of csrc/xdec.hip, which runs it as a phase of a persistent launch -- workgroup = (head, key split)
// of ONE image, `a2_smem` = a region of the launch's LDS, FRESH_DCTX = the context gradient was written by another CU of the same launch.
// Include inside namespace toist, after bf16_t / bf16x8_t / f32x4_t / pack2bf are visible.
#pragma once

namespace a2b {

typedef __attribute__((ext_vector_type(8))) short s16x8_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
typedef __attribute__((address_space(3))) s16x4_t* lds_v4;

constexpr float LOG2E = 1.4426950408889634f;
constexpr int A2_TS = 24;      // row stride (bf16) of a wave's [32 keys][16 queries] dS tile
constexpr int A2_KS = 40;      // row stride (bf16) of a wave's [32 keys][32 features] K tile (start-up transpose)
constexpr int A2_NG = 2;       // query groups per workgroup: tiles are dealt round-robin to A2_NG sets of four waves (two waves per SIMD)
constexpr size_t A2_BWD_LDS = (size_t)A2_NG * 2 * 32 * 32 * 2 * 2 + (size_t)A2_NG * 4 * 2 * 32 * A2_TS * 2 + (size_t)A2_NG * 2 * 32 * 4 + (size_t)A2_NG * 2 * 64 * 4 +
                              (size_t)A2_NG * 2 * 4 * 32 * 32 * 4;

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

template <bool FRESH_DCTX>
__device__ __forceinline__ void attn2_bwd_body(unsigned char* const a2_smem, const int split, const int bh,
                                               const bf16_t* __restrict__ q, int ldq, const bf16_t* __restrict__ kmat, int ldk,
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
    const int b = bh / H, h = bh - b * H;
    const int nkp = (Sk + 31) >> 5;                           // 32-key pairs of blocks in this head
    const int kp = split * 4 + w4;                       // this wave's pair
    const bool active = kp < nkp;
    const int nw = nkp - (int)split * 4 < 4 ? nkp - (int)split * 4 : 4;     // active waves of a group (>= 1)
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
    bf16_t* pout = (dq_part != nullptr ? dq_part + (size_t)split * part_stride : dq) + ((size_t)b * Sq + qrow0) * ldout + h * DH + (gtid & 7) * 4;
    const size_t sout = (size_t)A2_NG * QT * ldout;
    auto fetch_rows = [&](Rows& r, int q0_) {
        const int qi = q0_ + (gtid >> 3), ch = gtid & 7;
        r.d2 = u32x2_t{0, 0}; r.o2 = u32x2_t{0, 0}; r.q2 = u32x2_t{0, 0};
        r.ls = make_float2(0.f, 0.f);
        if (qi < Sq) {
            if (FRESH_DCTX) {      // written by another CU of this launch: sc1 (served by the L2, not by this CU's L1)
                const unsigned long long raw = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(pd), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                r.d2 = u32x2_t{(unsigned)raw, (unsigned)(raw >> 32)};
            } else {
                r.d2 = *reinterpret_cast<const u32x2_t*>(pd);
            }
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


}  // namespace a2b
