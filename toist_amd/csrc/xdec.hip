// XCD-resident decoder stack, forward: all decoder layers of /root/reference/models/transformer.py:225-267 (TransformerDecoder.forward) /
// :362-408 (TransformerDecoderLayer.forward_post) in ONE launch.  include/toist_hip.h (toist_xdec_fwd) states the plan; this file is it.
//
// Why: at batch 8 a decoder layer is 100 query rows per image -- 0.3 GFLOP -- and ran as 10 dependent launches of 6-12 us (77 us per
// layer forward, profiles/r04_timeline_graph_step.txt: the launches' fixed costs, not their flops).  Images never interact inside the
// decoder, the batch is 8 and the chip has 8 XCDs with a private L2 each, so one image is given to one XCD: its 32 workgroups pass
// activations through that L2 and meet at XCD-local barriers (0.84 us, profiles/r05_xcd_barrier.txt) instead of at kernel boundaries.
//
// Group formation is a runtime fact, not an assumption about the dispatcher: a workgroup reads HW_REG_XCC_ID and takes a ticket on that
// XCD's counter; the 32 ticket holders of one XCD share one physical L2, which is all the protocol below relies on:
//   producer: plain stores -> every wave `s_waitcnt vmcnt(0)` (the L2 has acknowledged them) -> __syncthreads -> one lane adds 1 to the
//             XCD's arrival counter (L2-scope atomic, no sc1)
//   consumer: one lane polls the counter with sc1 loads (L1 bypass) -> __syncthreads -> sc1 loads of the data (the reader's L1 may hold
//             stale lines of a buffer it read one layer earlier; sc1 reads are served by the L2)
// Weights, memory K / V, query_pos and the key-padding mask were written by earlier launches and are read with plain loads.
// Every spin is bounded: on expiry the status word is set, the workgroup stops waiting and the host raises (toist_amd.xdec.check_status).
//
// Arithmetic conventions are those of the per-op kernels, because their backward launches consume what this kernel saves:
//   attention     csrc/attn2.hip attn2_fwd_kernel: p = exp2((s - m) c), (row maximum of the raw dot products, 1 / row sum) in lse,
//                 one 32-bit pair hash per two keys (pair index = (row * round8(Sk) + key) >> 1)
//   sub-layers    csrc/tlayer.hip rowgemm_kernel LN_FWD: z = bf16(dropout(a W^T + b) + residual), statistics of the ROUNDED row,
//                 dropout hash of element m * 256 + n; linear1: dropout(relu(.)) hashed on element m * 2048 + n (csrc/gemm.hip epilogue)
#include "common.h"

#include <type_traits>

namespace toist {

namespace {

typedef __attribute__((ext_vector_type(8))) short s16x8_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
typedef __attribute__((address_space(3))) s16x4_t* lds_v4;
typedef __amdgpu_buffer_rsrc_t rsrc_t;

constexpr int XD = 256, XH = 8, XDH = 32, XFF = 2048;
constexpr int XWG = 32;            // workgroups per XCD = per image
constexpr int XNT = 512;           // threads per workgroup (8 waves: wave = head in the attention parts)
constexpr int XAS = XD + 8;        // row stride (bf16) of 256-wide operand rows in LDS
constexpr int XCS = XD + 8;        // row stride (f32) of the accumulator rows in LDS
constexpr int XW2S = 72;           // row stride (bf16) of the [256 n][64 hidden] linear2 slice in LDS
constexpr int XPR = 128;           // rows per (image, CU) slab of the linear2 partial sums
constexpr unsigned XOOB = 0x80000000u;      // a byte offset beyond num_records: the buffer load returns zeros without touching memory
constexpr float LOG2E = 1.4426950408889634f;
constexpr unsigned XSPIN_MAX = 1u << 20;    // ~0.3 s of polling: only a group that is not co-resident gets there

// LDS carve (bytes, every offset a multiple of 16)
constexpr int L_SX = 0;                                   // [4][XAS] bf16: attention context rows (A operand of the out projections)
constexpr int L_SY = L_SX + 4 * XAS * 2;                  // [4][XAS] bf16: norm1 output + query_pos (A operand of the cross-attention query projection)
constexpr int L_SQ = L_SY + 4 * XAS * 2;                  // [4][XAS] bf16: query rows of the attention in progress
constexpr int L_SC = L_SQ + 4 * XAS * 2;                  // [4][XCS] f32: accumulator rows
constexpr int L_DEADC = L_SC + 4 * XCS * 4;               // [512] u8: memory keys that are padding or beyond S
constexpr int L_DEADS = L_DEADC + 512;                    // [128] u8: query keys beyond Q
constexpr int L_INFO = L_DEADS + 128;                     // [16] u32
constexpr int L_SV = L_INFO + 64;                         // [8 waves][64 keys][32] bf16: V of the key block in progress, wave-private (32 KB; P7's sums need as much)
constexpr int L_RED = L_SV;                               // P7: [8][4][256] f32 (the V tiles are idle then)
constexpr int L_W1 = L_SV + 8 * 64 * XDH * 2;            // [64 hidden][XAS] bf16: this CU's rows of linear1
constexpr int L_W2 = L_W1 + 64 * XAS * 2;                 // [256 n][XW2S] bf16: this CU's columns of linear2
constexpr int L_TOTAL = L_W2 + XD * XW2S * 2;
static_assert(L_TOTAL <= 160 * 1024, "xdec: LDS budget");
static_assert(8 * 4 * XD * 4 <= 8 * 64 * XDH * 2, "xdec: the P7 sums alias the V tiles");

template <int N, typename F>
__device__ __forceinline__ void unrolled(F&& f) {
    if constexpr (N > 0) {
        unrolled<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

__device__ __forceinline__ rsrc_t mkrs(const void* p) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7ffffff0, 0x00020000); }
// 16 bytes at byte offset `off` of the buffer; FRESH = written by another CU of this launch: sc1 (served by the XCD's L2, not by this CU's L1)
template <bool FRESH>
__device__ __forceinline__ bf16x8_t ld16(rsrc_t r, unsigned off) {
    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, FRESH ? 16 : 0);
    return __builtin_bit_cast(bf16x8_t, v);
}
template <bool FRESH>
__device__ __forceinline__ uint4 ld16u(rsrc_t r, unsigned off) {
    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, FRESH ? 16 : 0);
    return make_uint4(v[0], v[1], v[2], v[3]);
}

__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 15u;
}

// attn2.hip's hash: one 32-bit value decides two adjacent keys of a score row
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
__device__ __forceinline__ float group16_sum(float v) {
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    v += __shfl_xor(v, 8, 64);
    return v;
}

// ---- XCD-local barrier ------------------------------------------------------------------------------------------------------------------
struct XSync {
    unsigned* arrive;      // this XCD's arrival counter
    unsigned* status;      // ctl[1023]
    unsigned* s_dead;      // LDS flag: a spin expired in this workgroup
    unsigned epoch;
};
__device__ __forceinline__ void xcd_barrier(XSync& sy) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's stores have reached the L2
    __syncthreads();
    sy.epoch += XWG;
    if (threadIdx.x == 0 && *sy.s_dead == 0u) {
        __hip_atomic_fetch_add(sy.arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);        // performed in the L2 (no sc1)
        unsigned spins = 0;
        while (__hip_atomic_load(sy.arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < sy.epoch) {   // global_load_dword sc1
            __builtin_amdgcn_s_sleep(1);
            if (++spins > XSPIN_MAX) {
                __hip_atomic_store(sy.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                *sy.s_dead = 1u;
                break;
            }
        }
    }
    __syncthreads();
}

// rows [b * Q, (b + 1) * Q) of the images b = xcd, xcd + 8, ... in `copies` stacked [rows][ld] bf16 tensors, columns 0 .. cols - 1 (cols % 8 == 0) := NaN
__device__ __forceinline__ void poison_rows(bf16_t* base, size_t ld, int cols, int copies, size_t copy_stride, int xcd, int B, int Q) {
    const uint4 nan8 = make_uint4(0x7FC07FC0u, 0x7FC07FC0u, 0x7FC07FC0u, 0x7FC07FC0u);
    const int per_row = cols >> 3;
    for (int b = xcd; b < B; b += 8)
        for (int c = 0; c < copies; ++c)
            for (int i = threadIdx.x; i < Q * per_row; i += XNT) {
                const int r = i / per_row, pc = i - r * per_row;
                *reinterpret_cast<uint4*>(base + c * copy_stride + ((size_t)b * Q + r) * ld + pc * 8) = nan8;
            }
}

__device__ __forceinline__ void xstamp(uint64_t* prof, int wg, int L, int layer, int phase) {
    if (prof != nullptr && threadIdx.x == 0) prof[((size_t)wg * L + layer) * 16 + phase] = wall_clock64();
}

// ---- weights of a 256 -> 256 projection as MFMA B fragments, straight from global memory --------------------------------------------------
// wave w owns output columns 32 w .. 32 w + 31 (two 16-column blocks); a lane (column c16, k group g) takes 32 contiguous bytes per 64-deep
// k-step (k = 64 s + 16 g .. + 15): the four k groups of a row read one whole 128-byte line, and the A fragments use the same k order.
struct WFrag {
    bf16x8_t w[2][8];
};
__device__ __forceinline__ void wload(WFrag& f, const void* W, int wave, int c16, int g) {
    const rsrc_t rs = mkrs(W);
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        const unsigned off = (unsigned)(((wave * 32 + nb * 16 + c16) * XD + 16 * g) * 2);
#pragma unroll
        for (int i = 0; i < 8; ++i) f.w[nb][i] = ld16<false>(rs, off + (unsigned)((64 * (i >> 1) + 8 * (i & 1)) * 2));
    }
}
// acc rows: the A operand's 16 MFMA rows are the 4 real rows replicated (row c16 & 3), so every lane group g holds rows 0 .. 3 in r
__device__ __forceinline__ void rgemm(const WFrag& f, const bf16_t* sA, int c16, int g, f32x4_t* acc) {
    acc[0] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    acc[1] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const bf16_t* a = sA + (c16 & 3) * XAS + 16 * g;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const bf16x8_t af = *reinterpret_cast<const bf16x8_t*>(a + 64 * (i >> 1) + 8 * (i & 1));
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, f.w[0][i], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, f.w[1][i], acc[1], 0, 0, 0);
    }
}
__device__ __forceinline__ void acc_to_lds(const f32x4_t* acc, float* sC, int wave, int c16, int g) {
    if (g == 0) {
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int r = 0; r < 4; ++r) sC[r * XCS + wave * 32 + nb * 16 + c16] = acc[nb][r];
    }
}


// ---- row epilogue: ONE WAVE PER ROW, a lane owns 4 consecutive columns (4 lane .. 4 lane + 3) ------------------------------------------------
// The operands that do not depend on the GEMM (bias, affine parameters, query_pos) are requested early (ln_prefetch) so that they are not queued
// behind a weight stream when the row is ready (vmcnt retires in order).
struct LnPre {
    float4 bias, gamma, beta;
    uint2 add;
};
__device__ __forceinline__ LnPre ln_prefetch(const float* bias, const float* gamma, const float* beta, const bf16_t* add_row, int lane) {
    LnPre o;
    o.bias = *reinterpret_cast<const float4*>(bias + 4 * lane);
    o.gamma = *reinterpret_cast<const float4*>(gamma + 4 * lane);
    o.beta = *reinterpret_cast<const float4*>(beta + 4 * lane);
    o.add = add_row != nullptr ? *reinterpret_cast<const uint2*>(add_row + 4 * lane) : make_uint2(0u, 0u);
    return o;
}
__device__ __forceinline__ void unpack4f(const uint2 u, float* v) {
    v[0] = __uint_as_float(u.x << 16);
    v[1] = __uint_as_float(u.x & 0xffff0000u);
    v[2] = __uint_as_float(u.y << 16);
    v[3] = __uint_as_float(u.y & 0xffff0000u);
}
struct LnOut {
    bf16_t* z;             // global row pointers (this wave's row), or nullptr
    bf16_t* y;
    bf16_t* y2;
    float* mean;
    float* rstd;
    bf16_t* sOut;          // LDS row: receives y2 (y when the row has no addend), or nullptr
};
// sCrow: the row's 256 accumulators (f32, LDS); resid (in): the residual row, (out): this LayerNorm's output row (f32 values of the bf16 numbers)
__device__ __forceinline__ void ln_row(const float* sCrow, const LnPre& o, float (&resid)[4], unsigned long long seed, float drop_p, float eps, size_t grow, bool live,
                                       bool has_add, const LnOut& out, int lane) {
    const float4 a = *reinterpret_cast<const float4*>(sCrow + 4 * lane);
    float v[4] = {a.x + o.bias.x, a.y + o.bias.y, a.z + o.bias.z, a.w + o.bias.w};
    if (drop_p > 0.f) {
        const unsigned thresh = (unsigned)(drop_p * 4294967296.0);
        const float dscale = 1.f / (1.f - drop_p);
        const unsigned long long idx = (unsigned long long)grow * XD + 4 * lane;
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = dropout_keep(seed, idx + q, thresh) ? v[q] * dscale : 0.f;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] += resid[q];
    const uint2 zp = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
    if (live && out.z != nullptr) *reinterpret_cast<uint2*>(out.z + 4 * lane) = zp;
    unpack4f(zp, v);                                     // the statistics are those of the ROUNDED row (what the backward pass reads)
    const float mean = wave_sum(v[0] + v[1] + v[2] + v[3]) * (1.f / XD);
    float qq = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) { const float d = v[q] - mean; qq += d * d; }
    const float rstd = rsqrtf(wave_sum(qq) * (1.f / XD) + eps);
    const float y0 = (v[0] - mean) * rstd * o.gamma.x + o.beta.x, y1 = (v[1] - mean) * rstd * o.gamma.y + o.beta.y;
    const float y2 = (v[2] - mean) * rstd * o.gamma.z + o.beta.z, y3 = (v[3] - mean) * rstd * o.gamma.w + o.beta.w;
    uint2 yo = make_uint2(pack2bf(y0, y1), pack2bf(y2, y3));
    if (live) *reinterpret_cast<uint2*>(out.y + 4 * lane) = yo;
    unpack4f(yo, resid);
    if (has_add) {
        float a4[4];
        unpack4f(o.add, a4);
        yo = make_uint2(pack2bf(resid[0] + a4[0], resid[1] + a4[1]), pack2bf(resid[2] + a4[2], resid[3] + a4[3]));
        if (live && out.y2 != nullptr) *reinterpret_cast<uint2*>(out.y2 + 4 * lane) = yo;
    }
    if (out.sOut != nullptr) *reinterpret_cast<uint2*>(out.sOut + 4 * lane) = yo;
    if (live && lane == 0) {
        *out.mean = mean;
        *out.rstd = rstd;
    }
}

// ---- attention of 4 query rows against Sk keys, one wave = one head (the block body is attn2_fwd_kernel's) -----------------------------
// qf: this lane's query fragment (query c16 & 3, features 8 g .. 8 g + 7 of the head); K / V: buffer + byte offset of (key 0, this head's first
// feature) and the row stride in bytes; FRESH = the buffer was written in this launch.  The loads of key block i + 1 are in flight while block i
// is computed; `between` runs once, right after the first block's loads have been issued (the caller's prefetch of the next weight stream).
// Writes the context rows into sX (LDS) and to global memory, (max, 1 / sum) to lse.
constexpr int XKB = 64;        // keys per block of the attention loop (128 as in attn2.hip keeps 64 more registers live next to the prefetched weight stream: spills)
struct KVBlock {
    bf16x8_t kf[XKB / 16];
    uint4 vv[XKB / 16];
};
template <bool FRESH>
__device__ __forceinline__ void kv_load(KVBlock& t, rsrc_t rsK, unsigned offK, rsrc_t rsV, unsigned offV, unsigned ldb, int Sk, int k0, int lane) {
    const int g = lane >> 4, c16 = lane & 15;
#pragma unroll
    for (int j = 0; j < XKB / 16; ++j) {
        const int key = k0 + j * 16 + c16;
        t.kf[j] = ld16<FRESH>(rsK, key < Sk ? offK + (unsigned)key * ldb + (unsigned)(g * 16) : XOOB);
    }
#pragma unroll
    for (int u = 0; u < XKB / 16; ++u) {
        const int pc = u * 64 + lane, key = k0 + (pc >> 2);
        t.vv[u] = ld16u<FRESH>(rsV, key < Sk ? offV + (unsigned)key * ldb + (unsigned)((pc & 3) * 16) : XOOB);
    }
}
template <bool FRESH, typename Between>
__device__ __forceinline__ void attn_rows(const bf16x8_t qf, rsrc_t rsK, unsigned offK, rsrc_t rsV, unsigned offV, unsigned ldb, int Sk, const unsigned char* sDead,
                                          bf16_t* sVw, int h, int lane, float c, float drop_p, unsigned long long seed, unsigned row, bool qlive, bf16_t* sX,
                                          bf16_t* ctx_row, float* lse_row, Between&& between) {
    constexpr int NJ = XKB / 16;
    const int g = lane >> 4, c16 = lane & 15;
    const bool dropping = drop_p > 0.f;
    const unsigned t16 = dropping ? (unsigned)(drop_p * 65536.0f + 0.5f) : 0u;
    const unsigned s0 = (unsigned)seed, s1 = (unsigned)(seed >> 32) ^ ((unsigned)seed * 0x9E3779B9u);
    const int ldp = (Sk + 7) & ~7;
    const unsigned pair_row = row * (unsigned)(ldp >> 1);
    float m = -INFINITY, l = 0.f;
    f32x4_t acc[2] = {f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}};
    KVBlock cur;
    kv_load<FRESH>(cur, rsK, offK, rsV, offV, ldb, Sk, 0, lane);
    between();
    for (int k0 = 0; k0 < Sk; k0 += XKB) {
#pragma unroll
        for (int u = 0; u < NJ; ++u) {
            const int pc = u * 64 + lane;
            *reinterpret_cast<uint4*>(sVw + (pc >> 2) * XDH + (pc & 3) * 8) = cur.vv[u];
        }
        f32x4_t s[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) s[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cur.kf[j], qf, f32x4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        if (k0 + XKB < Sk) kv_load<FRESH>(cur, rsK, offK, rsV, offV, ldb, Sk, k0 + XKB, lane);          // the next block's rows travel during this block's arithmetic
        const unsigned char dd = sDead[k0 + lane];
        if (__any(dd != 0)) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const unsigned dead4 = *reinterpret_cast<const unsigned*>(sDead + k0 + j * 16 + g * 4);
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if ((dead4 >> (8 * r)) & 0xffu) s[j][r] = -INFINITY;
            }
        }
        float bm = -INFINITY;
#pragma unroll
        for (int j = 0; j < NJ; ++j) bm = fmaxf(fmaxf(bm, fmaxf(s[j][0], s[j][1])), fmaxf(s[j][2], s[j][3]));
        bm = fmaxf(bm, __shfl_xor(bm, 16, 64));
        bm = fmaxf(bm, __shfl_xor(bm, 32, 64));
        const float mn = fmaxf(m, bm);
        const float msafe = (mn == -INFINITY) ? 0.f : mn;
        const float alpha = __builtin_amdgcn_exp2f((m - msafe) * c);
        m = mn;
        float psum = 0.f;
        unsigned pk[NJ][2];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            float p[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                p[r] = __builtin_amdgcn_exp2f((s[j][r] - msafe) * c);
                psum += p[r];
            }
            if (dropping) {
                const unsigned pair = pair_row + (unsigned)((k0 + j * 16 + 4 * g) >> 1);
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
#pragma unroll
        for (int cc = 0; cc < NJ / 2; ++cc) {
            const bf16x8_t pa = frag_of(pk[2 * cc][0], pk[2 * cc][1], pk[2 * cc + 1][0], pk[2 * cc + 1][1]);
            const int k_lo = 32 * cc + 4 * g + (c16 >> 2), k_hi = k_lo + 16;
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                const int col = nb * 16 + (c16 & 3) * 4;
                const bf16x8_t vb = tr_pair(sVw + k_lo * XDH + col, sVw + k_hi * XDH + col);
                acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vb, pa, acc[nb], 0, 0, 0);
            }
        }
    }
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.f / l;
    const float os = dropping ? inv / (1.f - drop_p) : inv;
    if (c16 < 4) {
        if (qlive && g == 0) *reinterpret_cast<float2*>(lse_row) = make_float2(m, inv);
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            const uint2 o = make_uint2(pack2bf(acc[nb][0] * os, acc[nb][1] * os), pack2bf(acc[nb][2] * os, acc[nb][3] * os));
            *reinterpret_cast<uint2*>(sX + c16 * XAS + h * XDH + nb * 16 + g * 4) = o;
            if (qlive) *reinterpret_cast<uint2*>(ctx_row + h * XDH + nb * 16 + g * 4) = o;
        }
    }
}

// pulls `bytes` (a multiple of 128) at `base` into this XCD's L2: thread idx of n touches every n-th line.  The sum keeps the loads alive.
__device__ __forceinline__ unsigned l2_touch(const void* base, size_t bytes, int idx, int n) {
    unsigned acc = 0;
    const unsigned* w = reinterpret_cast<const unsigned*>(base);
    for (size_t line = idx; line < bytes / 128; line += n) acc ^= w[line * 32];
    return acc;
}

}  // namespace

__global__ __launch_bounds__(XNT) void xdec_fwd_kernel(const toist_xdec_desc p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* const sX = reinterpret_cast<bf16_t*>(smem + L_SX);
    bf16_t* const sY = reinterpret_cast<bf16_t*>(smem + L_SY);
    bf16_t* const sQ = reinterpret_cast<bf16_t*>(smem + L_SQ);
    float* const sC = reinterpret_cast<float*>(smem + L_SC);
    unsigned char* const sDeadC = smem + L_DEADC;
    unsigned char* const sDeadS = smem + L_DEADS;
    unsigned* const sInfo = reinterpret_cast<unsigned*>(smem + L_INFO);
    float* const sRed = reinterpret_cast<float*>(smem + L_RED);
    bf16_t* const sW1 = reinterpret_cast<bf16_t*>(smem + L_W1);
    bf16_t* const sW2 = reinterpret_cast<bf16_t*>(smem + L_W2);
    const int tid0 = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
    bf16_t* const sVw = reinterpret_cast<bf16_t*>(smem + L_SV) + wave * (64 * XDH);

    // ---- group formation: the XCD is read from the hardware, the slot is a ticket of that XCD ----
    if (tid0 == 0) {
        const unsigned x = xcc_id() & 7u;
        sInfo[0] = x;
        sInfo[1] = __hip_atomic_fetch_add(p.ctl + x * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        sInfo[2] = 0u;
    }
    __syncthreads();
    const int xcd = __builtin_amdgcn_readfirstlane((int)sInfo[0]);
    const int slot = __builtin_amdgcn_readfirstlane((int)sInfo[1]);
    if (slot >= XWG || slot < p.test_absent) {      // a 33rd workgroup on one XCD takes no part (256 workgroups: 32 per XCD in every launch measured) -- some other XCD then
        if (tid0 == 0) __hip_atomic_store(p.ctl + (TOIST_XDEC_CTL_WORDS - 1), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // has 31 and its spins expire: say so at once
        return;                                       // (test_absent: the failure-path test makes `test_absent` workgroups per XCD leave like this)
    }
    XSync sy{p.ctl + 256 + xcd * 32, p.ctl + (TOIST_XDEC_CTL_WORDS - 1), sInfo + 2, 0u};

    const int Q = p.Q, S = p.S, M = p.B * p.Q;
    const int MT = (Q + 15) >> 4;                  // 16-row tiles of an image
    const int RB = (Q + 3) >> 2;                   // 4-row blocks of an image: block `slot` belongs to this CU
    const bool rowner = slot < RB;
    const int n_idle = XWG - RB;                   // CUs without rows: they pull the coming weights into the L2 while the row owners work
    const float cexp = 0.17677669529663687f * LOG2E;     // head dim 32: 1 / sqrt(32)
    const float drop_p = p.drop_p;
    const unsigned long long seed_add = p.seed_dev ? *p.seed_dev : 0ull;
    unsigned touched = 0;

    for (int b = xcd; b < p.B; b += 8) {
        int tid = tid0;
        asm volatile("" : "+v"(tid));
        int lane = tid & 63;
        __syncthreads();
        // masks of this image's keys
        for (int i = tid; i < 512; i += XNT) sDeadC[i] = (i >= S || (p.key_pad != nullptr && p.key_pad[(size_t)b * S + i])) ? 1 : 0;
        if (tid < 128) sDeadS[tid] = tid >= Q ? 1 : 0;
        const size_t row0 = (size_t)b * Q;             // first row of the image in the [B*Q, .] tensors
        const int my_row = 4 * slot + wave;            // waves 0 .. 3: the row this wave normalises
        const bool my_live = rowner && wave < 4 && my_row < Q;
        float resid[4] = {0.f, 0.f, 0.f, 0.f};          // waves 0 .. 3: the residual stream of the row (f32 of the bf16 values), 4 columns per lane
        if (my_live) unpack4f(*reinterpret_cast<const uint2*>(reinterpret_cast<const bf16_t*>(p.x0) + (row0 + my_row) * XD + 4 * lane), resid);
        if (b == xcd) {
            // layer 0's weights into this XCD's L2 (every CU takes a 32nd); later layers are fetched by the idle CUs one phase ahead
            const toist_xdec_layer& l0 = p.layer[0];
            const int idx = slot * XNT + tid, n = XWG * XNT;
            touched ^= l2_touch(l0.w_os, XD * XD * 2, idx, n) ^ l2_touch(l0.w_q, XD * XD * 2, idx, n) ^ l2_touch(l0.w_oc, XD * XD * 2, idx, n) ^
                       l2_touch(l0.w1, (size_t)XFF * XD * 2, idx, n) ^ l2_touch(l0.w2, (size_t)XFF * XD * 2, idx, n);
        }
        __syncthreads();

        for (int layer = 0; layer < p.L; ++layer) {
            // per-lane offsets are re-derived in every layer: hoisted out of the loop they are a few hundred live registers (128 spills)
            // (and again at the head of every phase: XD_REDERIVE)
            int g, c16;
#define XD_REDERIVE()                   \
    do {                                \
        asm volatile("" : "+v"(tid));   \
        lane = tid & 63;                \
        g = lane >> 4;                  \
        c16 = lane & 15;                \
    } while (0)
            XD_REDERIVE();
            const toist_xdec_layer& ly = p.layer[layer];
            const size_t lrow = (size_t)layer * M + row0;        // row offset of (layer, image) in the stacked outputs
            const bf16_t* const x_img = layer == 0 ? reinterpret_cast<const bf16_t*>(p.x0) + row0 * XD : reinterpret_cast<const bf16_t*>(p.y4) + (lrow - M) * XD;
            const bf16_t* const xe_img = layer == 0 ? reinterpret_cast<const bf16_t*>(p.qpos) + row0 * XD : reinterpret_cast<const bf16_t*>(p.y4e) + (lrow - M) * XD;
            bf16_t* const qkv_img = reinterpret_cast<bf16_t*>(p.qkv) + lrow * (3 * XD);
            const bf16_t* const qpos_img = reinterpret_cast<const bf16_t*>(p.qpos) + row0 * XD;
            const bool last = layer + 1 == p.L;

            const int wgid = xcd * XWG + slot;
            xstamp(p.prof, wgid, p.L, layer, 0);
            // ================= P1: q | k | v column tiles (16 columns each): CUs 0 .. 15 compute tiles slot and slot + 16 (q, k: both read x + query_pos,
            // one set of A fragments), CUs 16 .. 31 tile slot + 16 (v: reads x); wave = 16-row tile =================
            if (wave < MT) {
                const int mrow = wave * 16 + c16;
                const unsigned aoff = mrow < Q ? (unsigned)((mrow * XD + 16 * g) * 2) : XOOB;
                const rsrc_t rsW = mkrs(ly.w_in);
                const bool two = slot < 16;
                const rsrc_t rsA = mkrs(two ? xe_img : x_img);
                const int nt0 = two ? slot : slot + 16, nt1 = slot + 16;
                bf16x8_t af[8], wf[8], wf2[8];
                const unsigned woff = (unsigned)(((nt0 * 16 + c16) * XD + 16 * g) * 2), woff2 = (unsigned)(((nt1 * 16 + c16) * XD + 16 * g) * 2);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const unsigned ko = (unsigned)((64 * (i >> 1) + 8 * (i & 1)) * 2);
                    af[i] = layer == 0 ? ld16<false>(rsA, aoff + ko) : ld16<true>(rsA, aoff + ko);
                    wf[i] = ld16<false>(rsW, woff + ko);
                    wf2[i] = ld16<false>(rsW, two ? woff2 + ko : XOOB);
                }
                f32x4_t acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], af[i], acc, 0, 0, 0);       // [n = 4 g + r][m = c16]
                    acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf2[i], af[i], acc2, 0, 0, 0);
                }
                const float4 bb = *reinterpret_cast<const float4*>(ly.b_in + nt0 * 16 + 4 * g), b2 = *reinterpret_cast<const float4*>(ly.b_in + nt1 * 16 + 4 * g);
                if (mrow < Q) {
                    *reinterpret_cast<uint2*>(qkv_img + (size_t)mrow * (3 * XD) + nt0 * 16 + 4 * g) =
                        make_uint2(pack2bf(acc[0] + bb.x, acc[1] + bb.y), pack2bf(acc[2] + bb.z, acc[3] + bb.w));
                    if (two)
                        *reinterpret_cast<uint2*>(qkv_img + (size_t)mrow * (3 * XD) + nt1 * 16 + 4 * g) =
                            make_uint2(pack2bf(acc2[0] + b2.x, acc2[1] + b2.y), pack2bf(acc2[2] + b2.z, acc2[3] + b2.w));
                }
            }
            xstamp(p.prof, wgid, p.L, layer, 1);
            xcd_barrier(sy);
            xstamp(p.prof, wgid, p.L, layer, 2);

            // ================= A: the row owners: self-attention, norm1, cross-attention, norm3 for rows 4 slot .. 4 slot + 3 =================
            XD_REDERIVE();
            if (rowner) {
                const int q0 = 4 * slot;
                const int qi = q0 + (c16 & 3);
                const bool qlive = qi < Q;
                WFrag wf;
                LnPre pre;
                f32x4_t acc[2];
                {
                    const rsrc_t rsk = mkrs(qkv_img);
                    const bf16x8_t qf = ld16<true>(rsk, qlive ? (unsigned)((qi * (3 * XD) + wave * XDH + g * 8) * 2) : XOOB);
                    const unsigned long long seed = ly.seed[0] + seed_add;
                    const unsigned row = (unsigned)((b * XH + wave) * Q + (qlive ? qi : 0));
                    attn_rows<true>(qf, rsk, (unsigned)((XD + wave * XDH) * 2), rsk, (unsigned)((2 * XD + wave * XDH) * 2), (unsigned)(3 * XD * 2), Q, sDeadS, sVw, wave,
                                    lane, cexp, drop_p, seed, row, qlive, sX, reinterpret_cast<bf16_t*>(p.ctx_s) + (lrow + (qlive ? qi : 0)) * XD,
                                    p.lse_s + ((size_t)layer * p.B * XH * Q + (size_t)row) * 2, [&]() {
                                        // behind the K / V requests: norm1's operands, then the out-projection weights (they stream during the softmax)
                                        if (wave < 4) pre = ln_prefetch(ly.b_os, ly.g1, ly.be1, my_live ? qpos_img + (size_t)my_row * XD : nullptr, lane);
                                        wload(wf, ly.w_os, wave, c16, g);
                                    });
                }
                xstamp(p.prof, wgid, p.L, layer, 8);
                __syncthreads();                                   // sX complete (all heads)
                xstamp(p.prof, wgid, p.L, layer, 9);
                rgemm(wf, sX, c16, g, acc);
                acc_to_lds(acc, sC, wave, c16, g);
                __syncthreads();
                if (wave < 4) {
                    const LnOut o{reinterpret_cast<bf16_t*>(p.z1) + (lrow + my_row) * XD, reinterpret_cast<bf16_t*>(p.y1) + (lrow + my_row) * XD,
                                  reinterpret_cast<bf16_t*>(p.y1e) + (lrow + my_row) * XD, p.mean1 + lrow + my_row, p.rstd1 + lrow + my_row, sY + wave * XAS};
                    ln_row(sC + wave * XCS, pre, resid, ly.seed[1] + seed_add, drop_p, p.eps, row0 + my_row, my_live, true, o, lane);
                    pre = ln_prefetch(ly.b_oc, ly.g3, ly.be3, nullptr, lane);        // norm3's operands: ahead of the next weight streams
                }
                wload(wf, ly.w_q, wave, c16, g);
                __syncthreads();                                   // sY = norm1 output + query_pos
                xstamp(p.prof, wgid, p.L, layer, 10);
                rgemm(wf, sY, c16, g, acc);
                acc_to_lds(acc, sC, wave, c16, g);
                __syncthreads();
                if (tid < 128) {                                   // cross-attention queries: + bias, bf16 -> LDS rows and global
                    const int r = tid >> 5, pc = tid & 31;
                    float v8[8];
                    const float4 lo = *reinterpret_cast<const float4*>(sC + r * XCS + pc * 8), hi = *reinterpret_cast<const float4*>(sC + r * XCS + pc * 8 + 4);
                    const float4 b0 = *reinterpret_cast<const float4*>(ly.b_q + pc * 8), b1 = *reinterpret_cast<const float4*>(ly.b_q + pc * 8 + 4);
                    v8[0] = lo.x + b0.x; v8[1] = lo.y + b0.y; v8[2] = lo.z + b0.z; v8[3] = lo.w + b0.w;
                    v8[4] = hi.x + b1.x; v8[5] = hi.y + b1.y; v8[6] = hi.z + b1.z; v8[7] = hi.w + b1.w;
                    const uint4 o = pack8f(v8);
                    *reinterpret_cast<uint4*>(sQ + r * XAS + pc * 8) = o;
                    if (q0 + r < Q) *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(p.qc) + (lrow + q0 + r) * XD + pc * 8) = o;
                }
                __syncthreads();
                xstamp(p.prof, wgid, p.L, layer, 11);
                XD_REDERIVE();
                {
                    const bf16_t* kv_img = reinterpret_cast<const bf16_t*>(p.kv) + (size_t)b * S * p.ldkv;
                    const rsrc_t rsk = mkrs(kv_img);
                    const bf16x8_t qf = *reinterpret_cast<const bf16x8_t*>(sQ + (c16 & 3) * XAS + wave * XDH + g * 8);
                    const unsigned long long seed = ly.seed[2] + seed_add;
                    const unsigned row = (unsigned)((b * XH + wave) * Q + (qlive ? qi : 0));
                    attn_rows<false>(qf, rsk, (unsigned)((layer * 2 * XD + wave * XDH) * 2), rsk, (unsigned)((layer * 2 * XD + XD + wave * XDH) * 2),
                                     (unsigned)(p.ldkv * 2), S, sDeadC, sVw, wave, lane, cexp, drop_p, seed, row, qlive, sX,
                                     reinterpret_cast<bf16_t*>(p.ctx_c) + (lrow + (qlive ? qi : 0)) * XD,
                                     p.lse_c + ((size_t)layer * p.B * XH * Q + (size_t)row) * 2, [&]() { wload(wf, ly.w_oc, wave, c16, g); });
                }
                xstamp(p.prof, wgid, p.L, layer, 12);
                __syncthreads();
                xstamp(p.prof, wgid, p.L, layer, 13);
                rgemm(wf, sX, c16, g, acc);
                acc_to_lds(acc, sC, wave, c16, g);
                __syncthreads();
                if (wave < 4) {
                    const LnOut o{reinterpret_cast<bf16_t*>(p.z3) + (lrow + my_row) * XD, reinterpret_cast<bf16_t*>(p.y3) + (lrow + my_row) * XD, nullptr,
                                  p.mean3 + lrow + my_row, p.rstd3 + lrow + my_row, nullptr};
                    ln_row(sC + wave * XCS, pre, resid, ly.seed[3] + seed_add, drop_p, p.eps, row0 + my_row, my_live, false, o, lane);
                }
            } else {
                // idle CUs: this layer's FFN weights and the next layer's projections into the L2, a phase ahead of their readers
                const int idx = (slot - RB) * XNT + tid, n = n_idle * XNT;
                if (layer > 0) touched ^= l2_touch(ly.w1, (size_t)XFF * XD * 2, idx, n) ^ l2_touch(ly.w2, (size_t)XFF * XD * 2, idx, n);
                if (!last) {
                    const toist_xdec_layer& nx = p.layer[layer + 1];
                    touched ^= l2_touch(nx.w_in, 3 * XD * XD * 2, idx, n) ^ l2_touch(nx.w_os, XD * XD * 2, idx, n) ^ l2_touch(nx.w_q, XD * XD * 2, idx, n) ^
                               l2_touch(nx.w_oc, XD * XD * 2, idx, n);
                }
            }
            xstamp(p.prof, wgid, p.L, layer, 3);
            XD_REDERIVE();
            // this CU's slices of linear1 / linear2 -> LDS (requested before the barrier, they land while the group gathers)
            {
                uint4 w1v[4], w2v[4];
                const rsrc_t rs1 = mkrs(ly.w1), rs2 = mkrs(ly.w2);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int pc = tid + XNT * u;
                    w1v[u] = ld16u<false>(rs1, (unsigned)(((slot * 64 + (pc >> 5)) * XD + (pc & 31) * 8) * 2));
                    w2v[u] = ld16u<false>(rs2, (unsigned)(((pc >> 3) * XFF + slot * 64 + (pc & 7) * 8) * 2));
                }
                xcd_barrier(sy);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int pc = tid + XNT * u;
                    *reinterpret_cast<uint4*>(sW1 + (pc >> 5) * XAS + (pc & 31) * 8) = w1v[u];
                    *reinterpret_cast<uint4*>(sW2 + (pc >> 3) * XW2S + (pc & 7) * 8) = w2v[u];
                }
            }
            __syncthreads();
            xstamp(p.prof, wgid, p.L, layer, 4);

            // ================= P6: hidden units 64 slot .. 64 slot + 63: h = dropout(relu(y3 W1^T + b1)), partial sums of linear2 =================
            XD_REDERIVE();
            LnPre pre4;
            if (my_live) pre4 = ln_prefetch(ly.b2, ly.g4, ly.be4, last ? nullptr : qpos_img + (size_t)my_row * XD, lane);       // norm4's operands, a phase early
            if (wave < MT) {
                const int mrow = wave * 16 + c16;
                const bool mlive = mrow < Q;
                const rsrc_t rsy = mkrs(reinterpret_cast<const bf16_t*>(p.y3) + lrow * XD);
                const unsigned yoff = mlive ? (unsigned)((mrow * XD + 16 * g) * 2) : XOOB;
                bf16x8_t yf[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) yf[i] = ld16<true>(rsy, yoff + (unsigned)((64 * (i >> 1) + 8 * (i & 1)) * 2));
                const unsigned long long seed = ly.seed[4] + seed_add;
                const unsigned thresh = drop_p > 0.f ? (unsigned)(drop_p * 4294967296.0) : 0u;
                const float dscale = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
                unsigned pk[4][2];
                bf16_t* const h_row = reinterpret_cast<bf16_t*>(p.h) + (lrow + (mlive ? mrow : 0)) * XFF + slot * 64;
                if (p.prof != nullptr) {          // diagnostics: when have the y3 fragments arrived?
#pragma unroll
                    for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(yf[i]));
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    xstamp(p.prof, wgid, p.L, layer, 14);
                }
#pragma unroll
                for (int th = 0; th < 4; ++th) {
                    f32x4_t a4 = {0.f, 0.f, 0.f, 0.f};
                    const bf16_t* wrow = sW1 + (16 * th + c16) * XAS + 16 * g;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const bf16x8_t wfr = *reinterpret_cast<const bf16x8_t*>(wrow + 64 * (i >> 1) + 8 * (i & 1));
                        a4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfr, yf[i], a4, 0, 0, 0);        // [hidden 16 th + 4 g + r][m = c16]
                    }
                    const int hid = slot * 64 + 16 * th + 4 * g;
                    const float4 bb = *reinterpret_cast<const float4*>(ly.b1 + hid);
                    float hv[4] = {fmaxf(a4[0] + bb.x, 0.f), fmaxf(a4[1] + bb.y, 0.f), fmaxf(a4[2] + bb.z, 0.f), fmaxf(a4[3] + bb.w, 0.f)};
                    if (drop_p > 0.f) {
                        const unsigned long long idx = (unsigned long long)(row0 + mrow) * XFF + hid;
#pragma unroll
                        for (int r = 0; r < 4; ++r) hv[r] = dropout_keep(seed, idx + r, thresh) ? hv[r] * dscale : 0.f;
                    }
                    pk[th][0] = pack2bf(hv[0], hv[1]);
                    pk[th][1] = pack2bf(hv[2], hv[3]);
                    if (mlive) *reinterpret_cast<uint2*>(h_row + 16 * th + 4 * g) = make_uint2(pk[th][0], pk[th][1]);
                }
                // linear2 partial: k slots of block u: (g, i) <-> hidden 32 u + 4 g + i, (g, 4 + i) <-> hidden 32 u + 16 + 4 g + i
                const bf16x8_t pa0 = frag_of(pk[0][0], pk[0][1], pk[1][0], pk[1][1]), pa1 = frag_of(pk[2][0], pk[2][1], pk[3][0], pk[3][1]);
                bf16_t* const part_row = reinterpret_cast<bf16_t*>(p.part) + (((size_t)b * XWG + slot) * XPR + mrow) * XD;
                if (p.prof != nullptr) {
                    asm volatile("" : "+v"(pk[3][1]));
                    xstamp(p.prof, wgid, p.L, layer, 15);
                }
#pragma unroll
                for (int v = 0; v < 8; ++v) {
                    f32x4_t o[2];
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const int n = 32 * v + 8 * (c16 >> 2) + 4 * e + (c16 & 3);        // MFMA row 4 g' + r  <->  column 32 v + 8 g' + 4 e + r
                        const bf16_t* w2r = sW2 + n * XW2S + 4 * g;
                        const uint2 a0 = *reinterpret_cast<const uint2*>(w2r), a1 = *reinterpret_cast<const uint2*>(w2r + 16);
                        const uint2 b0 = *reinterpret_cast<const uint2*>(w2r + 32), b1 = *reinterpret_cast<const uint2*>(w2r + 48);
                        f32x4_t t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_of(a0.x, a0.y, a1.x, a1.y), pa0, f32x4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                        o[e] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_of(b0.x, b0.y, b1.x, b1.y), pa1, t, 0, 0, 0);
                    }
                    // lane (m = c16, g) holds columns 32 v + 8 g .. + 7
                    if (mlive)
                        *reinterpret_cast<uint4*>(part_row + 32 * v + 8 * g) =
                            make_uint4(pack2bf(o[0][0], o[0][1]), pack2bf(o[0][2], o[0][3]), pack2bf(o[1][0], o[1][1]), pack2bf(o[1][2], o[1][3]));
                }
            }
            xstamp(p.prof, wgid, p.L, layer, 5);
            xcd_barrier(sy);
            xstamp(p.prof, wgid, p.L, layer, 6);

            // ================= P7: the row owners add the 32 partial sums: + bias, dropout, residual, norm4 =================
            XD_REDERIVE();
            if (rowner) {
                {
                    const rsrc_t rsp = mkrs(reinterpret_cast<const bf16_t*>(p.part) + (size_t)b * XWG * XPR * XD);
                    const int r = lane >> 4, pl = lane & 15;
                    float sum[2][8];
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                        for (int q = 0; q < 8; ++q) sum[hh][q] = 0.f;
                    uint4 pv[4][2];
#pragma unroll
                    for (int u = 0; u < 4; ++u)
#pragma unroll
                        for (int hh = 0; hh < 2; ++hh)
                            pv[u][hh] = ld16u<true>(rsp, (unsigned)((((wave * 4 + u) * XPR + 4 * slot + r) * XD + (pl + 16 * hh) * 8) * 2));
#pragma unroll
                    for (int u = 0; u < 4; ++u)
#pragma unroll
                        for (int hh = 0; hh < 2; ++hh) {
                            float t8[8];
                            unpack8f(pv[u][hh], t8);
#pragma unroll
                            for (int q = 0; q < 8; ++q) sum[hh][q] += t8[q];
                        }
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh) {
                        float* dst = sRed + (wave * 4 + r) * XD + (pl + 16 * hh) * 8;
                        *reinterpret_cast<float4*>(dst) = make_float4(sum[hh][0], sum[hh][1], sum[hh][2], sum[hh][3]);
                        *reinterpret_cast<float4*>(dst + 4) = make_float4(sum[hh][4], sum[hh][5], sum[hh][6], sum[hh][7]);
                    }
                }
                __syncthreads();
                if (wave < 4) {                                     // wave = row: the eight waves' sums in a fixed order, then the row epilogue
                    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int w = 0; w < 8; ++w) {
                        const float4 s4 = *reinterpret_cast<const float4*>(sRed + (w * 4 + wave) * XD + 4 * lane);
                        t.x += s4.x; t.y += s4.y; t.z += s4.z; t.w += s4.w;
                    }
                    *reinterpret_cast<float4*>(sC + wave * XCS + 4 * lane) = t;      // (same lane reads it back: no barrier needed)
                    const LnOut o{reinterpret_cast<bf16_t*>(p.z4) + (lrow + my_row) * XD, reinterpret_cast<bf16_t*>(p.y4) + (lrow + my_row) * XD,
                                  last ? nullptr : reinterpret_cast<bf16_t*>(p.y4e) + (lrow + my_row) * XD, p.mean4 + lrow + my_row, p.rstd4 + lrow + my_row, nullptr};
                    ln_row(sC + wave * XCS, pre4, resid, ly.seed[5] + seed_add, drop_p, p.eps, row0 + my_row, my_live, !last, o, lane);
                }
            }
            xstamp(p.prof, wgid, p.L, layer, 7);
            if (layer + 1 < p.L || b + 8 < p.B) xcd_barrier(sy);
#undef XD_REDERIVE
        }
    }
    if (touched == 0x9E3779B9u) __hip_atomic_store(p.ctl + 1000, touched, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // keeps the L2 touches alive
    // A bounded spin expired in this workgroup (its group was not co-resident): once one member stops arriving every other member's next
    // wait expires too, and nothing the group wrote can be trusted.  Make that unmissable without a host read: every layer output of the
    // group's images becomes NaN, hence NaN logits / boxes / losses (the reference's finite-loss guard, engine.py:82-85, trips in the same step).
    __syncthreads();
    if (*sy.s_dead != 0u) poison_rows(reinterpret_cast<bf16_t*>(p.y4), (size_t)XD, XD, p.L, (size_t)M * XD, xcd, p.B, Q);
}


// =====================================================================================================================================================
// Backward (include/toist_hip.h: toist_xdec_bwd).  Same grouping and barriers as the forward launch; the two attention backward phases run the body
// of csrc/attn2.hip's key-owning kernel (attn2_bwd_body.h) with workgroup = (head, key split) of this XCD's image.
#include "attn2_bwd_body.h"

namespace {

// LDS carve of the backward launch
constexpr int B_SA = 0;                                   // [4][XAS] bf16: A rows of the row-local data-gradient GEMMs
constexpr int B_SB = B_SA + 4 * XAS * 2;                  // [4][XAS] bf16
constexpr int B_SA3 = B_SB + 4 * XAS * 2;                 // [4][3 * XD + 8] bf16: [dq | dk | dv] rows
constexpr int B_SC = B_SA3 + 4 * (3 * XD + 8) * 2;        // [4][XCS] f32 accumulator rows
constexpr int B_SG = B_SC + 4 * XCS * 4;                  // [2][4][XD] f32: v * xhat, v of the four rows (LayerNorm parameter gradients)
constexpr int B_INFO = B_SG + 2 * 4 * XD * 4;             // [16] u32
constexpr int B_U = B_INFO + 64;                          // phase-private region:
constexpr int XKT = 64 * 40;                              //   one wave's [64 k][32 n + 8] weight tile (bf16 elements)
constexpr int B_TILE = B_U;                               //   row-local phases: [8 waves][2][XKT] bf16 ...
constexpr int B_RED = B_TILE + 8 * 2 * XKT * 2;           //   ... + [8][4][XD] f32 partial folds
constexpr int B_W1 = B_U;                                 //   H: [64 hidden][XAS] bf16 ...
constexpr int B_W2 = B_W1 + 64 * XAS * 2;                 //   ... + [256 n][XW2S] bf16
constexpr int B_END_ROW = B_RED + 8 * 4 * XD * 4, B_END_H = B_W2 + XD * XW2S * 2, B_END_ATT = B_U + (int)a2b::A2_BWD_LDS;
constexpr int B_TOTAL = (B_END_ROW > B_END_H ? (B_END_ROW > B_END_ATT ? B_END_ROW : B_END_ATT) : (B_END_H > B_END_ATT ? B_END_H : B_END_ATT));
static_assert(B_TOTAL <= 160 * 1024, "xdec backward: LDS budget");
static_assert((B_U % 16) == 0, "xdec backward: alignment");

template <bool FRESH>
__device__ __forceinline__ uint2 ld8(rsrc_t r, unsigned off) {
    typedef __attribute__((ext_vector_type(2))) unsigned u2_t;
    const u2_t v = __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, FRESH ? 16 : 0);
    return make_uint2(v[0], v[1]);
}

// ---- row-local data gradient: acc (this wave's 32 output columns, the 4 rows replicated) = A[4][K] W[K][256], W = an nn.Linear weight read in place
// (its ROWS are the reduction index): 64-row chunks of the wave's 32 columns go through a wave-private LDS tile and come back transposed. ----
template <int K>
__device__ __forceinline__ void kgemm(const bf16_t* sA, int lda, const void* W, bf16_t* tiles, int wave, int lane, f32x4_t* acc) {
    const int g = lane >> 4, c16 = lane & 15;
    acc[0] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    acc[1] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const rsrc_t rs = mkrs(W);
    const unsigned lane_off = (unsigned)(((lane >> 2) * XD + wave * 32 + (lane & 3) * 8) * 2);       // row lane / 4 (+ 16 i), 16-byte piece lane % 4 of the wave's 64-byte row slice
    const int wr = (lane >> 2) * 40 + (lane & 3) * 8;
    const bf16_t* arow = sA + (c16 & 3) * lda + 4 * g;
    for (int n0 = 0; n0 < K; n0 += 256) {                       // 256 reduction rows = four 64-row chunks = 16 requests in flight per lane
        uint4 wv[4][4];
#pragma unroll
        for (int ch = 0; ch < 4; ++ch)
#pragma unroll
            for (int i = 0; i < 4; ++i) wv[ch][i] = ld16u<false>(rs, lane_off + (unsigned)((n0 + ch * 64 + 16 * i) * XD * 2));
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) {
            bf16_t* const tile = tiles + (ch & 1) * XKT;
#pragma unroll
            for (int i = 0; i < 4; ++i) *reinterpret_cast<uint4*>(tile + 16 * i * 40 + wr) = wv[ch][i];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int kk = n0 + ch * 64 + 32 * u;
                const uint2 a0 = *reinterpret_cast<const uint2*>(arow + kk), a1 = *reinterpret_cast<const uint2*>(arow + kk + 16);
                const bf16x8_t af = frag_of(a0.x, a0.y, a1.x, a1.y);
                const bf16_t* tb = tile + (32 * u + 4 * g + (c16 >> 2)) * 40 + (c16 & 3) * 4;
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) {
                    const bf16x8_t bfr = tr_pair(tb + nb * 16, tb + nb * 16 + 16 * 40);
                    acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bfr, acc[nb], 0, 0, 0);
                }
            }
        }
    }
}

// ---- LayerNorm backward of one row by one wave (a lane owns 4 columns): v = gradient of the LayerNorm output (f32) ----
// dz (out): the input gradient, rounded to bf16 like the tensor the per-op launch stores; the dropout-masked copy goes to global memory and (optionally) to LDS
__device__ __forceinline__ void lnb_row(const float (&v_in)[4], const bf16_t* z_row, float mu, float rs, const float* gamma, unsigned long long seed, float drop_p,
                                        size_t grow, bool live, float (&dz)[4], bf16_t* gb_row, bf16_t* sOutRow, float* sG, int row, int lane) {
    float z4[4], v[4];
    unpack4f(live ? *reinterpret_cast<const uint2*>(z_row + 4 * lane) : make_uint2(0u, 0u), z4);
    const float4 gm = *reinterpret_cast<const float4*>(gamma + 4 * lane);
    const float gam[4] = {gm.x, gm.y, gm.z, gm.w};
    float xh[4], gg[4], s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        v[q] = live ? v_in[q] : 0.f;
        xh[q] = (z4[q] - mu) * rs;
        gg[q] = v[q] * gam[q];
        s1 += gg[q];
        s2 += gg[q] * xh[q];
    }
    s1 = wave_sum(s1) * (1.f / XD);
    s2 = wave_sum(s2) * (1.f / XD);
    float o[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) o[q] = rs * (gg[q] - s1 - xh[q] * s2);
    const uint2 dzp = make_uint2(pack2bf(o[0], o[1]), pack2bf(o[2], o[3]));
    unpack4f(dzp, dz);
    uint2 gbp = dzp;
    if (drop_p > 0.f) {
        const unsigned thresh = (unsigned)(drop_p * 4294967296.0);
        const float dscale = 1.f / (1.f - drop_p);
        const unsigned long long idx = (unsigned long long)grow * XD + 4 * lane;
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q] = dropout_keep(seed, idx + q, thresh) ? o[q] * dscale : 0.f;
        gbp = make_uint2(pack2bf(o[0], o[1]), pack2bf(o[2], o[3]));
    }
    if (live) *reinterpret_cast<uint2*>(gb_row + 4 * lane) = gbp;
    if (sOutRow != nullptr) *reinterpret_cast<uint2*>(sOutRow + 4 * lane) = gbp;
    *reinterpret_cast<float4*>(sG + row * XD + 4 * lane) = make_float4(v[0] * xh[0], v[1] * xh[1], v[2] * xh[2], v[3] * xh[3]);
    *reinterpret_cast<float4*>(sG + (4 + row) * XD + 4 * lane) = make_float4(v[0], v[1], v[2], v[3]);
}

}  // namespace

__global__ __launch_bounds__(XNT) void xdec_bwd_kernel(const toist_xdec_bwd_desc p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* const sA = reinterpret_cast<bf16_t*>(smem + B_SA);
    bf16_t* const sB = reinterpret_cast<bf16_t*>(smem + B_SB);
    bf16_t* const sA3 = reinterpret_cast<bf16_t*>(smem + B_SA3);
    float* const sC = reinterpret_cast<float*>(smem + B_SC);
    float* const sG = reinterpret_cast<float*>(smem + B_SG);
    unsigned* const sInfo = reinterpret_cast<unsigned*>(smem + B_INFO);
    float* const sRed = reinterpret_cast<float*>(smem + B_RED);
    bf16_t* const sW1 = reinterpret_cast<bf16_t*>(smem + B_W1);
    bf16_t* const sW2 = reinterpret_cast<bf16_t*>(smem + B_W2);
    const int tid0 = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
    bf16_t* const tiles = reinterpret_cast<bf16_t*>(smem + B_TILE) + wave * (2 * XKT);

    if (tid0 == 0) {
        const unsigned x = xcc_id() & 7u;
        sInfo[0] = x;
        sInfo[1] = __hip_atomic_fetch_add(p.ctl + x * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        sInfo[2] = 0u;
    }
    __syncthreads();
    const int xcd = __builtin_amdgcn_readfirstlane((int)sInfo[0]);
    const int slot = __builtin_amdgcn_readfirstlane((int)sInfo[1]);
    if (slot >= XWG || slot < p.test_absent) {
        if (tid0 == 0) __hip_atomic_store(p.ctl + (TOIST_XDEC_CTL_WORDS - 1), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    XSync sy{p.ctl + 256 + xcd * 32, p.ctl + (TOIST_XDEC_CTL_WORDS - 1), sInfo + 2, 0u};

    const int Q = p.Q, S = p.S, M = p.B * p.Q;
    const int MT = (Q + 15) >> 4;
    const int RB = (Q + 3) >> 2;
    const bool rowner = slot < RB;
    const float drop_p = p.drop_p;
    const unsigned long long seed_add = p.seed_dev ? *p.seed_dev : 0ull;
    const int splits_c = ((S + 31) / 32 + 3) / 4, splits_s = ((Q + 31) / 32 + 3) / 4;      // toist_attn2_splits
    const int wgid = xcd * XWG + slot;

    for (int b = xcd; b < p.B; b += 8) {
        int tid = tid0;
        asm volatile("" : "+v"(tid));
        int lane = tid & 63;
        const size_t row0 = (size_t)b * Q;
        const int my_row = 4 * slot + wave;
        const bool my_live = rowner && wave < 4 && my_row < Q;
        const int blk = b * RB + slot;                         // this CU's row block in the LayerNorm partial sums
        const int nblk = p.B * RB;
        float gy[4] = {0.f, 0.f, 0.f, 0.f};                     // waves 0 .. 3: gradient of the current layer's output row (f32), 4 columns per lane
        float rdz[4] = {0.f, 0.f, 0.f, 0.f};                    // the residual gradient that travels down the layer (bf16-rounded values)
        if (my_live) unpack4f(*reinterpret_cast<const uint2*>(reinterpret_cast<const bf16_t*>(p.g_out) + ((size_t)(p.L - 1) * M + row0 + my_row) * XD + 4 * lane), gy);
        __syncthreads();

        for (int layer = p.L - 1; layer >= 0; --layer) {
            int g, c16;
#define XB_REDERIVE()                   \
    do {                                \
        asm volatile("" : "+v"(tid));   \
        lane = tid & 63;                \
        g = lane >> 4;                  \
        c16 = lane & 15;                \
    } while (0)
            XB_REDERIVE();
            const toist_xdec_bwd_layer& ly = p.layer[layer];
            const size_t lrow = (size_t)layer * M + row0;
            float* const lnp = p.ln_part + (size_t)layer * 3 * 2 * nblk * XD;          // [3 norms][2][nblk][256]
            bf16_t* const sink_img = reinterpret_cast<bf16_t*>(p.sink) + row0 * p.ldsink + layer * 4 * XD;
            bf16_t* const dctx_c = reinterpret_cast<bf16_t*>(p.dctx), * const dctx_s = dctx_c + (size_t)M * XD;
            // writes this CU's LayerNorm partial rows (sum over its 4 rows of v * xhat, v) of norm `which` (0: norm1, 1: norm3, 2: norm4); call after a barrier
            auto ln_partials = [&](int which) {
                if (tid < XD) {
                    float a = 0.f, c = 0.f;
#pragma unroll
                    for (int r = 0; r < 4; ++r) { a += sG[r * XD + tid]; c += sG[(4 + r) * XD + tid]; }
                    lnp[((size_t)(which * 2 + 0) * nblk + blk) * XD + tid] = a;
                    lnp[((size_t)(which * 2 + 1) * nblk + blk) * XD + tid] = c;
                }
            };

            xstamp(p.prof, wgid, p.L, layer, 0);
            // ================= R0: norm4 backward of this CU's rows =================
            if (rowner) {
                if (wave < 4) {
                    const float mu = my_live ? p.mean4[lrow + my_row] : 0.f, rs = my_live ? p.rstd4[lrow + my_row] : 0.f;
                    lnb_row(gy, reinterpret_cast<const bf16_t*>(p.z4) + (lrow + my_row) * XD, mu, rs, ly.g4, ly.seed[5] + seed_add, drop_p, row0 + my_row, my_live, rdz,
                            reinterpret_cast<bf16_t*>(p.gb4) + (lrow + my_row) * XD, nullptr, sG, wave, lane);
                }
                __syncthreads();
                ln_partials(2);
            }
            xstamp(p.prof, wgid, p.L, layer, 1);
            // this CU's slices of linear1 / linear2 -> registers before the barrier, -> LDS after it
            {
                uint4 w1v[4], w2v[4];
                const rsrc_t rs1 = mkrs(ly.w1), rs2 = mkrs(ly.w2);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int pc = tid + XNT * u;
                    w1v[u] = ld16u<false>(rs1, (unsigned)(((slot * 64 + (pc >> 5)) * XD + (pc & 31) * 8) * 2));
                    w2v[u] = ld16u<false>(rs2, (unsigned)(((pc >> 3) * XFF + slot * 64 + (pc & 7) * 8) * 2));
                }
                xcd_barrier(sy);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int pc = tid + XNT * u;
                    *reinterpret_cast<uint4*>(sW1 + (pc >> 5) * XAS + (pc & 31) * 8) = w1v[u];
                    *reinterpret_cast<uint4*>(sW2 + (pc >> 3) * XW2S + (pc & 7) * 8) = w2v[u];
                }
            }
            __syncthreads();
            xstamp(p.prof, wgid, p.L, layer, 2);

            // ================= H: hidden units 64 slot .. + 63: dh = (gb4 W2) where h > 0, partial sums of dh W1 =================
            XB_REDERIVE();
            if (wave < MT) {
                const int mrow = wave * 16 + c16;
                const bool mlive = mrow < Q;
                const rsrc_t rsg = mkrs(reinterpret_cast<const bf16_t*>(p.gb4) + lrow * XD);
                const unsigned goff = mlive ? (unsigned)((mrow * XD + 4 * g) * 2) : XOOB;
                bf16x8_t gbf[8];          // k slots of block j: (g, i) <-> n = 32 j + 4 g + i, (g, 4 + i) <-> n = 32 j + 16 + 4 g + i
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const uint2 lo = ld8<true>(rsg, goff + (unsigned)(64 * j)), hi = ld8<true>(rsg, goff + (unsigned)(64 * j + 32));
                    gbf[j] = frag_of(lo.x, lo.y, hi.x, hi.y);
                }
                const float alpha = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
                const bf16_t* const h_row = reinterpret_cast<const bf16_t*>(p.h) + (lrow + (mlive ? mrow : 0)) * XFF + slot * 64;
                bf16_t* const dh_row = reinterpret_cast<bf16_t*>(p.dh) + (lrow + (mlive ? mrow : 0)) * XFF + slot * 64;
                unsigned pk[4][2];
#pragma unroll
                for (int th = 0; th < 4; ++th) {
                    f32x4_t a4 = {0.f, 0.f, 0.f, 0.f};
                    const bf16_t* wt = sW2 + (4 * g + (c16 >> 2)) * XW2S + 16 * th + (c16 & 3) * 4;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const bf16x8_t wfr = tr_pair(wt + 32 * j * XW2S, wt + (32 * j + 16) * XW2S);     // lane = hidden 16 th + c16
                        a4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfr, gbf[j], a4, 0, 0, 0);          // [hidden 16 th + 4 g + r][m = c16]
                    }
                    float hv[4];
                    unpack4f(mlive ? *reinterpret_cast<const uint2*>(h_row + 16 * th + 4 * g) : make_uint2(0u, 0u), hv);
                    float dv4[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) dv4[r] = hv[r] > 0.f ? a4[r] * alpha : 0.f;
                    pk[th][0] = pack2bf(dv4[0], dv4[1]);
                    pk[th][1] = pack2bf(dv4[2], dv4[3]);
                    if (mlive) *reinterpret_cast<uint2*>(dh_row + 16 * th + 4 * g) = make_uint2(pk[th][0], pk[th][1]);
                }
                const bf16x8_t pa0 = frag_of(pk[0][0], pk[0][1], pk[1][0], pk[1][1]), pa1 = frag_of(pk[2][0], pk[2][1], pk[3][0], pk[3][1]);
                bf16_t* const part_row = reinterpret_cast<bf16_t*>(p.part) + (((size_t)b * XWG + slot) * XPR + mrow) * XD;
#pragma unroll
                for (int v = 0; v < 8; ++v) {
                    f32x4_t o[2];
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        // result lane L holds output column 32 v + 8 (L >> 2) + 4 e + (L & 3): the ADDRESS lane X supplies the 4-column chunk 32 v + 8 (X & 3) + 4 e
                        const bf16_t* w1t = sW1 + (4 * g + (c16 >> 2)) * XAS + 32 * v + 8 * (c16 & 3) + 4 * e;
                        const bf16x8_t f0 = tr_pair(w1t, w1t + 16 * XAS), f1 = tr_pair(w1t + 32 * XAS, w1t + 48 * XAS);
                        const f32x4_t t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f0, pa0, f32x4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                        o[e] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f1, pa1, t, 0, 0, 0);
                    }
                    if (mlive)
                        *reinterpret_cast<uint4*>(part_row + 32 * v + 8 * g) =
                            make_uint4(pack2bf(o[0][0], o[0][1]), pack2bf(o[0][2], o[0][3]), pack2bf(o[1][0], o[1][1]), pack2bf(o[1][2], o[1][3]));
                }
            }
            xstamp(p.prof, wgid, p.L, layer, 3);
            xcd_barrier(sy);
            xstamp(p.prof, wgid, p.L, layer, 4);

            // ================= C: fold the partials, norm3 backward, gradient of the cross-attention context =================
            XB_REDERIVE();
            if (rowner) {
                {
                    const rsrc_t rsp = mkrs(reinterpret_cast<const bf16_t*>(p.part) + (size_t)b * XWG * XPR * XD);
                    const int r = lane >> 4, pl = lane & 15;
                    float sum[2][8];
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                        for (int q = 0; q < 8; ++q) sum[hh][q] = 0.f;
                    uint4 pv[4][2];
#pragma unroll
                    for (int u = 0; u < 4; ++u)
#pragma unroll
                        for (int hh = 0; hh < 2; ++hh)
                            pv[u][hh] = ld16u<true>(rsp, (unsigned)((((wave * 4 + u) * XPR + 4 * slot + r) * XD + (pl + 16 * hh) * 8) * 2));
#pragma unroll
                    for (int u = 0; u < 4; ++u)
#pragma unroll
                        for (int hh = 0; hh < 2; ++hh) {
                            float t8[8];
                            unpack8f(pv[u][hh], t8);
#pragma unroll
                            for (int q = 0; q < 8; ++q) sum[hh][q] += t8[q];
                        }
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh) {
                        float* dst = sRed + (wave * 4 + r) * XD + (pl + 16 * hh) * 8;
                        *reinterpret_cast<float4*>(dst) = make_float4(sum[hh][0], sum[hh][1], sum[hh][2], sum[hh][3]);
                        *reinterpret_cast<float4*>(dst + 4) = make_float4(sum[hh][4], sum[hh][5], sum[hh][6], sum[hh][7]);
                    }
                }
                __syncthreads();
                if (wave < 4) {
                    float v[4] = {rdz[0], rdz[1], rdz[2], rdz[3]};              // the residual branch: norm4's input gradient
#pragma unroll
                    for (int w = 0; w < 8; ++w) {
                        const float4 s4 = *reinterpret_cast<const float4*>(sRed + (w * 4 + wave) * XD + 4 * lane);
                        v[0] += s4.x; v[1] += s4.y; v[2] += s4.z; v[3] += s4.w;
                    }
                    const float mu = my_live ? p.mean3[lrow + my_row] : 0.f, rs = my_live ? p.rstd3[lrow + my_row] : 0.f;
                    lnb_row(v, reinterpret_cast<const bf16_t*>(p.z3) + (lrow + my_row) * XD, mu, rs, ly.g3, ly.seed[3] + seed_add, drop_p, row0 + my_row, my_live, rdz,
                            reinterpret_cast<bf16_t*>(p.go3) + (lrow + my_row) * XD, sA + wave * XAS, sG, wave, lane);
                }
                __syncthreads();
                ln_partials(1);
                f32x4_t acc[2];
                kgemm<XD>(sA, XAS, ly.w_oc, tiles, wave, lane, acc);
                acc_to_lds(acc, sC, wave, c16, g);
                __syncthreads();
                if (tid < 128) {
                    const int r = tid >> 5, pc = tid & 31;
                    float v8[8];
                    const float4 lo = *reinterpret_cast<const float4*>(sC + r * XCS + pc * 8), hi = *reinterpret_cast<const float4*>(sC + r * XCS + pc * 8 + 4);
                    v8[0] = lo.x; v8[1] = lo.y; v8[2] = lo.z; v8[3] = lo.w; v8[4] = hi.x; v8[5] = hi.y; v8[6] = hi.z; v8[7] = hi.w;
                    if (4 * slot + r < Q) *reinterpret_cast<uint4*>(dctx_c + (row0 + 4 * slot + r) * XD + pc * 8) = pack8f(v8);
                }
            }
            xstamp(p.prof, wgid, p.L, layer, 5);
            xcd_barrier(sy);
            xstamp(p.prof, wgid, p.L, layer, 6);

            // ================= D: cross-attention backward, CU = (head slot / 4, key split slot % 4) =================
            if ((slot & 3) < splits_c) {
                const bf16_t* const kvb = reinterpret_cast<const bf16_t*>(p.kv) + layer * 2 * XD;
                bf16_t* const dkvb = reinterpret_cast<bf16_t*>(p.dkv) + layer * 2 * XD;
                a2b::attn2_bwd_body<true>(smem + B_U, slot & 3, b * XH + (slot >> 2), reinterpret_cast<const bf16_t*>(p.qc) + (size_t)layer * M * XD, XD, kvb, p.ldkv, kvb + XD,
                                          p.ldkv, reinterpret_cast<const bf16_t*>(p.ctx_c) + (size_t)layer * M * XD, XD, dctx_c, XD,
                                          p.lse_c + (size_t)layer * p.B * XH * Q * 2, p.key_pad, XH, Q, S, (S + 7) & ~7, 0.17677669529663687f, drop_p, ly.seed[2],
                                          reinterpret_cast<const unsigned long long*>(p.seed_dev), reinterpret_cast<bf16_t*>(p.sink) + layer * 4 * XD + 3 * XD, p.ldsink, dkvb,
                                          p.lddkv, dkvb + XD, p.lddkv, splits_c > 1 ? reinterpret_cast<bf16_t*>(p.dq_part) : nullptr, (long long)M * XD);
            }
            xstamp(p.prof, wgid, p.L, layer, 7);
            xcd_barrier(sy);
            xstamp(p.prof, wgid, p.L, layer, 8);

            // ================= E: fold dq, x W_q + residual, norm1 backward, gradient of the self-attention context =================
            XB_REDERIVE();
            if (rowner) {
                if (wave < 4) {
                    uint2 dq2;
                    if (splits_c > 1) {
                        const rsrc_t rsq = mkrs(reinterpret_cast<const bf16_t*>(p.dq_part));
                        float f4[4] = {0.f, 0.f, 0.f, 0.f};
                        for (int sp = 0; sp < splits_c; ++sp) {
                            float t4[4];
                            unpack4f(ld8<true>(rsq, my_live ? (unsigned)((((size_t)sp * M + row0 + my_row) * XD + 4 * lane) * 2) : XOOB), t4);
#pragma unroll
                            for (int q = 0; q < 4; ++q) f4[q] += t4[q];
                        }
                        dq2 = make_uint2(pack2bf(f4[0], f4[1]), pack2bf(f4[2], f4[3]));
                        if (my_live) *reinterpret_cast<uint2*>(sink_img + (size_t)my_row * p.ldsink + 3 * XD + 4 * lane) = dq2;     // the folded rows: dy of the query projection's weight gradient
                    } else {
                        const rsrc_t rsq = mkrs(sink_img);
                        dq2 = ld8<true>(rsq, my_live ? (unsigned)(((size_t)my_row * p.ldsink + 3 * XD + 4 * lane) * 2) : XOOB);
                    }
                    *reinterpret_cast<uint2*>(sA + wave * XAS + 4 * lane) = dq2;
                }
                __syncthreads();
                f32x4_t acc[2];
                kgemm<XD>(sA, XAS, ly.w_q, tiles, wave, lane, acc);
                acc_to_lds(acc, sC, wave, c16, g);
                __syncthreads();
                if (wave < 4) {
                    const float4 a4 = *reinterpret_cast<const float4*>(sC + wave * XCS + 4 * lane);
                    const float v[4] = {a4.x + rdz[0], a4.y + rdz[1], a4.z + rdz[2], a4.w + rdz[3]};          // + norm3's input gradient (the residual branch)
                    const float mu = my_live ? p.mean1[lrow + my_row] : 0.f, rs = my_live ? p.rstd1[lrow + my_row] : 0.f;
                    lnb_row(v, reinterpret_cast<const bf16_t*>(p.z1) + (lrow + my_row) * XD, mu, rs, ly.g1, ly.seed[1] + seed_add, drop_p, row0 + my_row, my_live, rdz,
                            reinterpret_cast<bf16_t*>(p.go1) + (lrow + my_row) * XD, sB + wave * XAS, sG, wave, lane);
                }
                __syncthreads();
                ln_partials(0);
                kgemm<XD>(sB, XAS, ly.w_os, tiles, wave, lane, acc);
                acc_to_lds(acc, sC, wave, c16, g);
                __syncthreads();
                if (tid < 128) {
                    const int r = tid >> 5, pc = tid & 31;
                    float v8[8];
                    const float4 lo = *reinterpret_cast<const float4*>(sC + r * XCS + pc * 8), hi = *reinterpret_cast<const float4*>(sC + r * XCS + pc * 8 + 4);
                    v8[0] = lo.x; v8[1] = lo.y; v8[2] = lo.z; v8[3] = lo.w; v8[4] = hi.x; v8[5] = hi.y; v8[6] = hi.z; v8[7] = hi.w;
                    if (4 * slot + r < Q) *reinterpret_cast<uint4*>(dctx_s + (row0 + 4 * slot + r) * XD + pc * 8) = pack8f(v8);
                }
            }
            xstamp(p.prof, wgid, p.L, layer, 9);
            xcd_barrier(sy);
            xstamp(p.prof, wgid, p.L, layer, 10);

            // ================= F: self-attention backward, CU = (head slot / 4, key split slot % 4) =================
            if ((slot & 3) < splits_s) {
                const bf16_t* const qkvb = reinterpret_cast<const bf16_t*>(p.qkv) + (size_t)layer * M * 3 * XD;
                bf16_t* const sk = reinterpret_cast<bf16_t*>(p.sink) + layer * 4 * XD;
                a2b::attn2_bwd_body<true>(smem + B_U, slot & 3, b * XH + (slot >> 2), qkvb, 3 * XD, qkvb + XD, 3 * XD, qkvb + 2 * XD, 3 * XD,
                                          reinterpret_cast<const bf16_t*>(p.ctx_s) + (size_t)layer * M * XD, XD, dctx_s, XD, p.lse_s + (size_t)layer * p.B * XH * Q * 2, nullptr,
                                          XH, Q, Q, (Q + 7) & ~7, 0.17677669529663687f, drop_p, ly.seed[0], reinterpret_cast<const unsigned long long*>(p.seed_dev), sk, p.ldsink,
                                          sk + XD, p.ldsink, sk + 2 * XD, p.ldsink, splits_s > 1 ? reinterpret_cast<bf16_t*>(p.dq_part) : nullptr, (long long)M * XD);
            }
            xstamp(p.prof, wgid, p.L, layer, 11);
            xcd_barrier(sy);
            xstamp(p.prof, wgid, p.L, layer, 12);

            // ================= G: [dq | dk | dv] W_in + residual gradient + the final norm's share of the layer below -> the next R0's input =================
            XB_REDERIVE();
            if (rowner && layer > 0) {
                if (tid < 384) {
                    const int r = tid / 96, pc = tid - r * 96;
                    const rsrc_t rsd = mkrs(sink_img);
                    uint4 v;
                    if (splits_s > 1 && pc < 32) {        // dq of a self-attention with several key splits arrives as shares (Q > 128 is outside the launch's limits: kept for completeness)
                        v = make_uint4(0u, 0u, 0u, 0u);
                    } else {
                        v = ld16u<true>(rsd, 4 * slot + r < Q ? (unsigned)(((size_t)(4 * slot + r) * p.ldsink + pc * 8) * 2) : XOOB);
                    }
                    *reinterpret_cast<uint4*>(sA3 + r * (3 * XD + 8) + pc * 8) = v;
                }
                __syncthreads();
                f32x4_t acc[2];
                kgemm<3 * XD>(sA3, 3 * XD + 8, ly.w_in, tiles, wave, lane, acc);
                acc_to_lds(acc, sC, wave, c16, g);
                __syncthreads();
                if (wave < 4) {
                    const float4 a4 = *reinterpret_cast<const float4*>(sC + wave * XCS + 4 * lane);
                    float go[4] = {0.f, 0.f, 0.f, 0.f};
                    if (my_live) unpack4f(*reinterpret_cast<const uint2*>(reinterpret_cast<const bf16_t*>(p.g_out) + ((size_t)(layer - 1) * M + row0 + my_row) * XD + 4 * lane), go);
                    gy[0] = a4.x + rdz[0] + go[0];
                    gy[1] = a4.y + rdz[1] + go[1];
                    gy[2] = a4.z + rdz[2] + go[2];
                    gy[3] = a4.w + rdz[3] + go[3];
                }
                __syncthreads();
            }
            xstamp(p.prof, wgid, p.L, layer, 13);
#undef XB_REDERIVE
        }
        if (b + 8 < p.B) xcd_barrier(sy);
    }
    // as in the forward launch: an expired spin turns the gradients every consumer reads (dq | dk | dv of the self-attention and dq of the cross-attention
    // of every layer: the query_pos gradient and the in_proj weight gradients) into NaN, so the gradient norm and the next loss are NaN
    __syncthreads();
    if (*sy.s_dead != 0u) poison_rows(reinterpret_cast<bf16_t*>(p.sink), (size_t)p.ldsink, p.L * 4 * XD, 1, 0, xcd, p.B, Q);
}

}  // namespace toist

using namespace toist;

extern "C" int toist_xdec_supported(int B, int Q, int S, int L) {
    if (B <= 0 || Q <= 0 || Q > 128 || S <= 0 || S > 512 || L <= 0 || L > TOIST_XDEC_MAX_LAYERS) return 0;
    static std::atomic<int> cached{-1};
    int ok = cached.load(std::memory_order_acquire);
    if (ok < 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        ok = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount == 256) ? 1 : 0;   // 8 XCDs x 32 CUs
        cached.store(ok, std::memory_order_release);
    }
    return ok;
}

extern "C" int toist_xdec_fwd(const toist_xdec_desc* d, void* stream) {
    TOIST_REQUIRE(d != nullptr, "toist_xdec_fwd: null descriptor");
    TOIST_REQUIRE(toist_xdec_supported(d->B, d->Q, d->S, d->L), "toist_xdec_fwd: unsupported shape B=%d Q=%d S=%d L=%d (Q <= 128, S <= 512, L <= 8, 256 CUs)", d->B, d->Q,
                  d->S, d->L);
    TOIST_REQUIRE(d->ff == XFF, "toist_xdec_fwd: dim_feedforward %d: the launch is compiled for linear1 [%d, 256] / linear2 [256, %d] (use the per-op path)", d->ff, XFF, XFF);
    TOIST_REQUIRE(d->test_absent >= 0 && d->test_absent <= XWG, "toist_xdec_fwd: bad test_absent");
    TOIST_REQUIRE(d->x0 && d->qpos && d->kv && d->qkv && d->ctx_s && d->lse_s && d->z1 && d->y1 && d->y1e && d->mean1 && d->rstd1 && d->qc && d->ctx_c && d->lse_c &&
                      d->z3 && d->y3 && d->mean3 && d->rstd3 && d->h && d->z4 && d->y4 && d->y4e && d->mean4 && d->rstd4 && d->part && d->ctl,
                  "toist_xdec_fwd: every buffer of the descriptor is required");
    TOIST_REQUIRE((d->ldkv % 8) == 0 && d->ldkv >= d->L * 2 * XD, "toist_xdec_fwd: ldkv %d must be a multiple of 8 and cover L * 512 columns", d->ldkv);
    TOIST_REQUIRE((long long)d->B * d->S * d->ldkv * 2 < 0x7ffffff0ll && (long long)d->L * d->B * d->Q * XFF * 2 < 0x7ffffff0ll, "toist_xdec_fwd: buffers beyond 2 GB");
    TOIST_REQUIRE((long long)d->B * XH * d->Q * ((d->S + 7) / 8 * 8) < (1ll << 32), "toist_xdec_fwd: dropout element index beyond 2^32");
    TOIST_REQUIRE(d->drop_p >= 0.f && d->drop_p < 1.f, "toist_xdec_fwd: bad dropout p");
    for (int l = 0; l < d->L; ++l) {
        const toist_xdec_layer& y = d->layer[l];
        TOIST_REQUIRE(y.w_in && y.b_in && y.w_os && y.b_os && y.g1 && y.be1 && y.w_q && y.b_q && y.w_oc && y.b_oc && y.g3 && y.be3 && y.w1 && y.b1 && y.w2 && y.b2 &&
                          y.g4 && y.be4,
                      "toist_xdec_fwd: layer %d: every parameter pointer is required", l);
        TOIST_REQUIRE(((((size_t)y.w_in) | ((size_t)y.w_os) | ((size_t)y.w_q) | ((size_t)y.w_oc) | ((size_t)y.w1) | ((size_t)y.w2) | ((size_t)y.b_in) | ((size_t)y.b_os) |
                        ((size_t)y.b_q) | ((size_t)y.b_oc) | ((size_t)y.b1) | ((size_t)y.b2) | ((size_t)y.g1) | ((size_t)y.be1) | ((size_t)y.g3) | ((size_t)y.be3) |
                        ((size_t)y.g4) | ((size_t)y.be4)) & 15) == 0,
                      "toist_xdec_fwd: layer %d: parameter pointers must be 16-byte aligned", l);
    }
    hipStream_t st = (hipStream_t)stream;
    static std::atomic<unsigned long long> done{0};
    if (!lds_attr_once_flag(done, [] { return hipFuncSetAttribute((const void*)xdec_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess; })) {
        set_last_error("toist_xdec_fwd: cannot raise the dynamic LDS limit");
        return TOIST_EHIP;
    }
    if (hipMemsetAsync(d->ctl, 0, (TOIST_XDEC_CTL_WORDS - 1) * sizeof(uint32_t), st)      /* the status word is sticky */ != hipSuccess) {
        set_last_error("toist_xdec_fwd: hipMemsetAsync of the control words failed");
        return TOIST_EHIP;
    }
    hipLaunchKernelGGL(xdec_fwd_kernel, dim3(8 * XWG), dim3(XNT), L_TOTAL, st, *d);
    return check_launch("toist_xdec_fwd");
}

extern "C" int toist_xdec_bwd(const toist_xdec_bwd_desc* d, void* stream) {
    TOIST_REQUIRE(d != nullptr, "toist_xdec_bwd: null descriptor");
    TOIST_REQUIRE(toist_xdec_supported(d->B, d->Q, d->S, d->L), "toist_xdec_bwd: unsupported shape B=%d Q=%d S=%d L=%d", d->B, d->Q, d->S, d->L);
    TOIST_REQUIRE(d->ff == XFF, "toist_xdec_bwd: dim_feedforward %d: the launch is compiled for %d hidden units (use the per-op path)", d->ff, XFF);
    TOIST_REQUIRE(d->test_absent >= 0 && d->test_absent <= XWG, "toist_xdec_bwd: bad test_absent");
    TOIST_REQUIRE(d->kv && d->qkv && d->ctx_s && d->lse_s && d->z1 && d->mean1 && d->rstd1 && d->qc && d->ctx_c && d->lse_c && d->z3 && d->mean3 && d->rstd3 && d->h && d->z4 &&
                      d->mean4 && d->rstd4 && d->g_out && d->gb4 && d->dh && d->go3 && d->go1 && d->sink && d->dkv && d->ln_part && d->dctx && d->part && d->dq_part && d->ctl,
                  "toist_xdec_bwd: every buffer of the descriptor is required");
    TOIST_REQUIRE((d->ldkv % 8) == 0 && d->ldkv >= d->L * 2 * XD && (d->lddkv % 8) == 0 && d->lddkv >= d->L * 2 * XD && (d->ldsink % 8) == 0 && d->ldsink >= d->L * 4 * XD,
                  "toist_xdec_bwd: ldkv / lddkv / ldsink must be multiples of 8 and cover L * 512 / L * 512 / L * 1024 columns");
    TOIST_REQUIRE((long long)d->B * d->S * d->ldkv * 2 < 0x7ffffff0ll && (long long)d->L * d->B * d->Q * XFF * 2 < 0x7ffffff0ll && (long long)d->B * d->Q * d->ldsink * 2 < 0x7ffffff0ll,
                  "toist_xdec_bwd: buffers beyond 2 GB");
    TOIST_REQUIRE(d->drop_p >= 0.f && d->drop_p < 1.f, "toist_xdec_bwd: bad dropout p");
    for (int l = 0; l < d->L; ++l) {
        const toist_xdec_bwd_layer& y = d->layer[l];
        TOIST_REQUIRE(y.w_in && y.w_os && y.w_q && y.w_oc && y.w1 && y.w2 && y.g1 && y.g3 && y.g4, "toist_xdec_bwd: layer %d: every parameter pointer is required", l);
        TOIST_REQUIRE(((((size_t)y.w_in) | ((size_t)y.w_os) | ((size_t)y.w_q) | ((size_t)y.w_oc) | ((size_t)y.w1) | ((size_t)y.w2) | ((size_t)y.g1) | ((size_t)y.g3) | ((size_t)y.g4)) & 15) == 0,
                      "toist_xdec_bwd: layer %d: parameter pointers must be 16-byte aligned", l);
    }
    hipStream_t st = (hipStream_t)stream;
    static std::atomic<unsigned long long> done{0};
    if (!lds_attr_once_flag(done, [] { return hipFuncSetAttribute((const void*)xdec_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess; })) {
        set_last_error("toist_xdec_bwd: cannot raise the dynamic LDS limit");
        return TOIST_EHIP;
    }
    if (hipMemsetAsync(d->ctl, 0, (TOIST_XDEC_CTL_WORDS - 1) * sizeof(uint32_t), st) != hipSuccess) {
        set_last_error("toist_xdec_bwd: hipMemsetAsync of the control words failed");
        return TOIST_EHIP;
    }
    hipLaunchKernelGGL(xdec_bwd_kernel, dim3(8 * XWG), dim3(XNT), B_TOTAL, st, *d);
    return check_launch("toist_xdec_bwd");
}
