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
constexpr int L_SV = L_INFO + 64;                         // [8 waves][128 keys][32] bf16: V of the key block in progress, wave-private
constexpr int L_RED = L_SV;                               // P7: [8][4][256] f32 (the V tiles are idle then)
constexpr int L_W1 = L_SV + 8 * 128 * XDH * 2;            // [64 hidden][XAS] bf16: this CU's rows of linear1
constexpr int L_W2 = L_W1 + 64 * XAS * 2;                 // [256 n][XW2S] bf16: this CU's columns of linear2
constexpr int L_TOTAL = L_W2 + XD * XW2S * 2;
static_assert(L_TOTAL <= 160 * 1024, "xdec: LDS budget");
static_assert(8 * 4 * XD * 4 <= 8 * 128 * XDH * 2, "xdec: the P7 sums alias the V tiles");

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

// ---- row epilogue of one wave: lane = (row lane >> 4, column pieces lane & 15 and (lane & 15) + 16), 8 columns per piece ----------------
// v (in)  : accumulator rows in LDS;  resid (in): the residual rows (f32 values of the bf16 tensor), (out): this LayerNorm's output rows
struct LnArgs {
    const float* bias;
    const float* gamma;
    const float* beta;
    const bf16_t* add;     // rows of query_pos (global, image rows), or nullptr
    bf16_t* z;             // global outputs, already offset to the image's first row
    bf16_t* y;
    bf16_t* y2;            // or nullptr
    float* mean;
    float* rstd;
    bf16_t* sOut;          // LDS [4][XAS]: receives y2 (or y when there is no add), or nullptr
    unsigned long long seed;
    float drop_p, eps;
};
__device__ __forceinline__ void ln_rows(const float* sC, const LnArgs& a, float (&resid)[2][8], int lane, int row_img, bool live, size_t grow) {
    const int r = lane >> 4, pl = lane & 15;
    float v[2][8];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int c0 = (pl + 16 * h) * 8;
        const float4 lo = *reinterpret_cast<const float4*>(sC + r * XCS + c0), hi = *reinterpret_cast<const float4*>(sC + r * XCS + c0 + 4);
        const float4 b0 = *reinterpret_cast<const float4*>(a.bias + c0), b1 = *reinterpret_cast<const float4*>(a.bias + c0 + 4);
        v[h][0] = lo.x + b0.x; v[h][1] = lo.y + b0.y; v[h][2] = lo.z + b0.z; v[h][3] = lo.w + b0.w;
        v[h][4] = hi.x + b1.x; v[h][5] = hi.y + b1.y; v[h][6] = hi.z + b1.z; v[h][7] = hi.w + b1.w;
    }
    if (a.drop_p > 0.f) {
        const unsigned thresh = (unsigned)(a.drop_p * 4294967296.0);
        const float dscale = 1.f / (1.f - a.drop_p);
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const unsigned long long idx = (unsigned long long)grow * XD + (pl + 16 * h) * 8 + q;
                v[h][q] = dropout_keep(a.seed, idx, thresh) ? v[h][q] * dscale : 0.f;
            }
    }
    float s = 0.f;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int q = 0; q < 8; ++q) v[h][q] += resid[h][q];
        const uint4 zp = pack8f(v[h]);
        if (live) *reinterpret_cast<uint4*>(a.z + (size_t)row_img * XD + (pl + 16 * h) * 8) = zp;
        unpack8f(zp, v[h]);
#pragma unroll
        for (int q = 0; q < 8; ++q) s += v[h][q];
    }
    const float mean = group16_sum(s) * (1.f / XD);
    float qq = 0.f;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int q = 0; q < 8; ++q) { const float d = v[h][q] - mean; qq += d * d; }
    const float rstd = rsqrtf(group16_sum(qq) * (1.f / XD) + a.eps);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int c0 = (pl + 16 * h) * 8;
        float gm[8], bt[8], o[8];
        *reinterpret_cast<float4*>(gm) = *reinterpret_cast<const float4*>(a.gamma + c0);
        *reinterpret_cast<float4*>(gm + 4) = *reinterpret_cast<const float4*>(a.gamma + c0 + 4);
        *reinterpret_cast<float4*>(bt) = *reinterpret_cast<const float4*>(a.beta + c0);
        *reinterpret_cast<float4*>(bt + 4) = *reinterpret_cast<const float4*>(a.beta + c0 + 4);
#pragma unroll
        for (int q = 0; q < 8; ++q) o[q] = (v[h][q] - mean) * rstd * gm[q] + bt[q];
        uint4 yo = pack8f(o);
        if (live) *reinterpret_cast<uint4*>(a.y + (size_t)row_img * XD + c0) = yo;
        unpack8f(yo, resid[h]);
        if (a.add != nullptr) {
            float a8[8], y8[8];
            unpack8f(live ? *reinterpret_cast<const uint4*>(a.add + (size_t)row_img * XD + c0) : make_uint4(0, 0, 0, 0), a8);
#pragma unroll
            for (int q = 0; q < 8; ++q) y8[q] = resid[h][q] + a8[q];
            yo = pack8f(y8);
            if (live && a.y2 != nullptr) *reinterpret_cast<uint4*>(a.y2 + (size_t)row_img * XD + c0) = yo;
        }
        if (a.sOut != nullptr) *reinterpret_cast<uint4*>(a.sOut + r * XAS + c0) = yo;
    }
    if (live && pl == 0) {
        a.mean[row_img] = mean;
        a.rstd[row_img] = rstd;
    }
}

// ---- attention of 4 query rows against Sk keys, one wave = one head (the loop body is attn2_fwd_kernel's) ------------------------------
// sQ: LDS rows [4][XAS] of the projected queries (all heads); K / V: buffer + byte offset of (key 0, this head's first feature) and the row
// stride in bytes; FRESH = the buffer was written in this launch.  Writes ctx rows into sX (LDS) and to global memory, (max, 1 / sum) to lse.
template <bool FRESH>
__device__ __forceinline__ void attn_rows(const bf16_t* sQ, rsrc_t rsK, unsigned offK, rsrc_t rsV, unsigned offV, unsigned ldb, int Sk,
                                          const unsigned char* sDead, bf16_t* sVw, int h, int lane, float c, float drop_p, unsigned long long seed,
                                          unsigned row, bool qlive, bf16_t* sX, bf16_t* ctx_row, float* lse_row) {
    const int g = lane >> 4, c16 = lane & 15;
    const bool dropping = drop_p > 0.f;
    const unsigned t16 = dropping ? (unsigned)(drop_p * 65536.0f + 0.5f) : 0u;
    const unsigned s0 = (unsigned)seed, s1 = (unsigned)(seed >> 32) ^ ((unsigned)seed * 0x9E3779B9u);
    const int ldp = (Sk + 7) & ~7;
    const unsigned pair_row = row * (unsigned)(ldp >> 1);
    const bf16x8_t qf = *reinterpret_cast<const bf16x8_t*>(sQ + (c16 & 3) * XAS + h * XDH + g * 8);
    float m = -INFINITY, l = 0.f;
    f32x4_t acc[2] = {f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}};
    for (int k0 = 0; k0 < Sk; k0 += 128) {
        // K fragments (lane = key 16 j + c16, features 8 g .. 8 g + 7) and the V rows of the block (16-byte pieces -> wave-private LDS tile)
        bf16x8_t kf[8];
        uint4 vv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int key = k0 + j * 16 + c16;
            kf[j] = ld16<FRESH>(rsK, key < Sk ? offK + (unsigned)key * ldb + (unsigned)(g * 16) : XOOB);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int pc = u * 64 + lane, key = k0 + (pc >> 2);
            vv[u] = ld16u<FRESH>(rsV, key < Sk ? offV + (unsigned)key * ldb + (unsigned)((pc & 3) * 16) : XOOB);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int pc = u * 64 + lane;
            *reinterpret_cast<uint4*>(sVw + (pc >> 2) * XDH + (pc & 3) * 8) = vv[u];
        }
        f32x4_t s[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) s[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[j], qf, f32x4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        const unsigned short dd = *reinterpret_cast<const unsigned short*>(sDead + k0 + lane * 2);
        if (__any(dd != 0)) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const unsigned dead4 = *reinterpret_cast<const unsigned*>(sDead + k0 + j * 16 + g * 4);
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
        const float msafe = (mn == -INFINITY) ? 0.f : mn;
        const float alpha = __builtin_amdgcn_exp2f((m - msafe) * c);
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
        for (int cc = 0; cc < 4; ++cc) {
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
    bf16_t* const sVw = reinterpret_cast<bf16_t*>(smem + L_SV) + wave * (128 * XDH);

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
    if (slot >= XWG) return;          // a 33rd workgroup on one XCD takes no part (256 workgroups: 32 per XCD in every launch measured)
    XSync sy{p.ctl + 256 + xcd * 32, p.ctl + (TOIST_XDEC_CTL_WORDS - 1), sInfo + 2, 0u};

    const int Q = p.Q, S = p.S, M = p.B * p.Q;
    const int MT = (Q + 15) >> 4;                  // 16-row tiles of an image
    const int RB = (Q + 3) >> 2;                   // 4-row blocks of an image: block `slot` belongs to this CU
    const bool rowner = slot < RB;
    const float cexp = 0.17677669529663687f * LOG2E;     // head dim 32: 1 / sqrt(32)
    const float drop_p = p.drop_p;
    const unsigned long long seed_add = p.seed_dev ? *p.seed_dev : 0ull;

    for (int b = xcd; b < p.B; b += 8) {
        int tid = tid0;
        asm volatile("" : "+v"(tid));
        int lane = tid & 63;
        __syncthreads();
        // masks of this image's keys
        for (int i = tid; i < 512; i += XNT) sDeadC[i] = (i >= S || (p.key_pad != nullptr && p.key_pad[(size_t)b * S + i])) ? 1 : 0;
        if (tid < 128) sDeadS[tid] = tid >= Q ? 1 : 0;
        const size_t row0 = (size_t)b * Q;             // first row of the image in the [B*Q, .] tensors
        const int my_row = 4 * slot + (lane >> 4);     // wave 0's row in the row epilogues
        const bool my_live = rowner && my_row < Q;
        float resid[2][8];                              // wave 0: the residual stream of this CU's rows (f32 of the bf16 values)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
            for (int q = 0; q < 8; ++q) resid[hh][q] = 0.f;
        if (wave == 0 && my_live) {
#pragma unroll
            for (int hh = 0; hh < 2; ++hh)
                unpack8f(*reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(p.x0) + (row0 + my_row) * XD + ((lane & 15) + 16 * hh) * 8), resid[hh]);
        }
        __syncthreads();

        for (int layer = 0; layer < p.L; ++layer) {
            // per-lane offsets are re-derived in every layer: hoisted out of the loop they are a few hundred live registers (128 spills)
            asm volatile("" : "+v"(tid));
            lane = tid & 63;
            const int g = lane >> 4, c16 = lane & 15;
            const toist_xdec_layer& ly = p.layer[layer];
            const size_t lrow = (size_t)layer * M + row0;        // row offset of (layer, image) in the stacked outputs
            const bf16_t* const x_img = layer == 0 ? reinterpret_cast<const bf16_t*>(p.x0) + row0 * XD : reinterpret_cast<const bf16_t*>(p.y4) + (lrow - M) * XD;
            const bf16_t* const xe_img = layer == 0 ? reinterpret_cast<const bf16_t*>(p.qpos) + row0 * XD : reinterpret_cast<const bf16_t*>(p.y4e) + (lrow - M) * XD;
            bf16_t* const qkv_img = reinterpret_cast<bf16_t*>(p.qkv) + lrow * (3 * XD);

            // ================= P1: q | k | v column tiles: CU `slot` computes 16-column tiles slot and slot + 32, wave = 16-row tile =================
            if (wave < MT) {
                const int mrow = wave * 16 + c16;
                const unsigned aoff = mrow < Q ? (unsigned)((mrow * XD + 16 * g) * 2) : XOOB;
                const rsrc_t rsW = mkrs(ly.w_in);
                for (int nt = slot; nt < 48; nt += XWG) {
                    const rsrc_t rsA = mkrs(nt < 32 ? xe_img : x_img);        // q, k from x + query_pos; v from x
                    const unsigned woff = (unsigned)(((nt * 16 + c16) * XD + 16 * g) * 2);
                    bf16x8_t af[8], wf[8];
                    if (layer == 0) {
#pragma unroll
                        for (int i = 0; i < 8; ++i) af[i] = ld16<false>(rsA, aoff + (unsigned)((64 * (i >> 1) + 8 * (i & 1)) * 2));
                    } else {
#pragma unroll
                        for (int i = 0; i < 8; ++i) af[i] = ld16<true>(rsA, aoff + (unsigned)((64 * (i >> 1) + 8 * (i & 1)) * 2));
                    }
#pragma unroll
                    for (int i = 0; i < 8; ++i) wf[i] = ld16<false>(rsW, woff + (unsigned)((64 * (i >> 1) + 8 * (i & 1)) * 2));
                    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int i = 0; i < 8; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], af[i], acc, 0, 0, 0);   // [n = 4 g + r][m = c16]
                    const float4 bb = *reinterpret_cast<const float4*>(ly.b_in + nt * 16 + 4 * g);
                    if (mrow < Q)
                        *reinterpret_cast<uint2*>(qkv_img + (size_t)mrow * (3 * XD) + nt * 16 + 4 * g) =
                            make_uint2(pack2bf(acc[0] + bb.x, acc[1] + bb.y), pack2bf(acc[2] + bb.z, acc[3] + bb.w));
                }
            }
            xcd_barrier(sy);

            // ================= A: the row owners: self-attention, norm1, cross-attention, norm3 for rows 4 slot .. 4 slot + 3 =================
            if (rowner) {
                const int q0 = 4 * slot;
                const int qi = q0 + (c16 & 3);
                const bool qlive = qi < Q;
                WFrag wf;
                // queries of the self-attention: q columns of qkv (all heads) -> LDS rows
                if (tid < 128) {
                    const int r = tid >> 5, pc = tid & 31;
                    const rsrc_t rsq = mkrs(qkv_img);
                    const uint4 v = ld16u<true>(rsq, q0 + r < Q ? (unsigned)(((q0 + r) * (3 * XD) + pc * 8) * 2) : XOOB);
                    *reinterpret_cast<uint4*>(sQ + r * XAS + pc * 8) = v;
                }
                __syncthreads();
                {
                    const rsrc_t rsk = mkrs(qkv_img);
                    const unsigned long long seed = ly.seed[0] + seed_add;
                    const unsigned row = (unsigned)((b * XH + wave) * Q + (qlive ? qi : 0));
                    attn_rows<true>(sQ, rsk, (unsigned)((XD + wave * XDH) * 2), rsk, (unsigned)((2 * XD + wave * XDH) * 2), (unsigned)(3 * XD * 2), Q, sDeadS,
                                    sVw, wave, lane, cexp, drop_p, seed, row, qlive, sX,
                                    reinterpret_cast<bf16_t*>(p.ctx_s) + (lrow + (qlive ? qi : 0)) * XD,
                                    p.lse_s + ((size_t)layer * p.B * XH * Q + (size_t)row) * 2);
                }
                wload(wf, ly.w_os, wave, c16, g);
                __syncthreads();                                   // sX complete (all heads)
                f32x4_t acc[2];
                rgemm(wf, sX, c16, g, acc);
                acc_to_lds(acc, sC, wave, c16, g);
                wload(wf, ly.w_q, wave, c16, g);                    // streams while wave 0 normalises
                __syncthreads();
                if (wave == 0) {
                    LnArgs a{ly.b_os, ly.g1, ly.be1, reinterpret_cast<const bf16_t*>(p.qpos) + row0 * XD,
                             reinterpret_cast<bf16_t*>(p.z1) + lrow * XD, reinterpret_cast<bf16_t*>(p.y1) + lrow * XD, reinterpret_cast<bf16_t*>(p.y1e) + lrow * XD,
                             p.mean1 + lrow, p.rstd1 + lrow, sY, ly.seed[1] + seed_add, drop_p, p.eps};
                    ln_rows(sC, a, resid, lane, my_row, my_live, row0 + my_row);
                }
                __syncthreads();                                   // sY = norm1 output + query_pos
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
                {
                    const bf16_t* kv_img = reinterpret_cast<const bf16_t*>(p.kv) + (size_t)b * S * p.ldkv;
                    const rsrc_t rsk = mkrs(kv_img);
                    const unsigned long long seed = ly.seed[2] + seed_add;
                    const unsigned row = (unsigned)((b * XH + wave) * Q + (qlive ? qi : 0));
                    attn_rows<false>(sQ, rsk, (unsigned)((layer * 2 * XD + wave * XDH) * 2), rsk, (unsigned)((layer * 2 * XD + XD + wave * XDH) * 2),
                                     (unsigned)(p.ldkv * 2), S, sDeadC, sVw, wave, lane, cexp, drop_p, seed, row, qlive, sX,
                                     reinterpret_cast<bf16_t*>(p.ctx_c) + (lrow + (qlive ? qi : 0)) * XD,
                                     p.lse_c + ((size_t)layer * p.B * XH * Q + (size_t)row) * 2);
                }
                wload(wf, ly.w_oc, wave, c16, g);
                __syncthreads();
                rgemm(wf, sX, c16, g, acc);
                acc_to_lds(acc, sC, wave, c16, g);
                __syncthreads();
                if (wave == 0) {
                    LnArgs a{ly.b_oc, ly.g3, ly.be3, nullptr,
                             reinterpret_cast<bf16_t*>(p.z3) + lrow * XD, reinterpret_cast<bf16_t*>(p.y3) + lrow * XD, nullptr,
                             p.mean3 + lrow, p.rstd3 + lrow, nullptr, ly.seed[3] + seed_add, drop_p, p.eps};
                    ln_rows(sC, a, resid, lane, my_row, my_live, row0 + my_row);
                }
            }
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

            // ================= P6: hidden units 64 slot .. 64 slot + 63: h = dropout(relu(y3 W1^T + b1)), partial sums of linear2 =================
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
            xcd_barrier(sy);

            // ================= P7: the row owners add the 32 partial sums: + bias, dropout, residual, norm4 =================
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
                if (wave == 0) {
                    const int r = lane >> 4, pl = lane & 15;
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh) {
                        float t[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int w = 0; w < 8; ++w) {
                            const float* src = sRed + (w * 4 + r) * XD + (pl + 16 * hh) * 8;
                            const float4 lo = *reinterpret_cast<const float4*>(src), hi = *reinterpret_cast<const float4*>(src + 4);
                            t[0] += lo.x; t[1] += lo.y; t[2] += lo.z; t[3] += lo.w; t[4] += hi.x; t[5] += hi.y; t[6] += hi.z; t[7] += hi.w;
                        }
                        float* dst = sC + r * XCS + (pl + 16 * hh) * 8;
                        *reinterpret_cast<float4*>(dst) = make_float4(t[0], t[1], t[2], t[3]);
                        *reinterpret_cast<float4*>(dst + 4) = make_float4(t[4], t[5], t[6], t[7]);
                    }
                    const bool last = layer + 1 == p.L;
                    LnArgs a{ly.b2, ly.g4, ly.be4, last ? nullptr : reinterpret_cast<const bf16_t*>(p.qpos) + row0 * XD,
                             reinterpret_cast<bf16_t*>(p.z4) + lrow * XD, reinterpret_cast<bf16_t*>(p.y4) + lrow * XD,
                             last ? nullptr : reinterpret_cast<bf16_t*>(p.y4e) + lrow * XD, p.mean4 + lrow, p.rstd4 + lrow, nullptr, ly.seed[5] + seed_add, drop_p, p.eps};
                    ln_rows(sC, a, resid, lane, my_row, my_live, row0 + my_row);
                }
            }
            if (layer + 1 < p.L || b + 8 < p.B) xcd_barrier(sy);
        }
    }
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
