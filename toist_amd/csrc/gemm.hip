// bf16 MFMA GEMM / implicit-GEMM convolution family for gfx950 (v_mfma_f32_16x16x32_bf16).
//
// One kernel template covers every dense contraction of the TOIST hot path:
//   * nn.Linear forward / dgrad / wgrad of the encoder-decoder, RoBERTa and the heads
//     (/root/reference/models/transformer.py:273-304,362-408; mdetr.py:420-433),
//   * the batched QK^T / PV products of attention,
//   * ResNet-101 convolutions as im2col-free implicit GEMM over NHWC activations, FrozenBatchNorm2d
//     (/root/reference/models/backbone.py:48-58) folded into the per-channel scale/shift epilogue,
//     including transposed-gather dgrad and pixel-reduction wgrad (backbone.py:64-66 trains layer2-4).
//
// Structure: 256 threads = 4 waves (2x2), tile BMxBNx32, global -> registers -> LDS staging with the
// next tile's loads in flight under the MFMAs, fp32 accumulation.  Operands whose reduction index is
// NOT the contiguous one (wgrad, PV, dgrad from the forward weight layout) are staged k-major and
// turned into MFMA fragments by the LDS transpose read ds_read_b64_tr_b16.  The MFMA is issued as
// D^T = B * A^T so each lane owns 4 consecutive output columns -> 8-byte packed bf16 stores.
#include <type_traits>

#include "common.h"

namespace toist {

// BK (template parameter) = k extent of one staged tile; k-contiguous LDS tiles use a row pitch of
// BK + 8 elements (80 / 144 B) so the 16-lane groups of a ds_read_b128 spread over the banks.

__device__ __forceinline__ float gelu_f(float v) { return 0.5f * v * (1.f + erff(v * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad_f(float a) {
    return 0.5f * (1.f + erff(a * 0.70710678118654752f)) + a * 0.3989422804014327f * __expf(-0.5f * a * a);
}

// One 1x4 output fragment: v = acc*alpha*rscale[m] -> *scale[n] + shift[n] -> dropout -> + res -> act -> dropout -> store.
// Written with compile-time element indices only (no break/continue) so accumulators stay in registers.
__device__ __forceinline__ void epilogue_frag(const toist_gemm& p, const f32x4_t a, const int m, const int n, const int bz,
                                              const long long coff) {
    const toist_epilogue& e = p.epi;
    const int N = p.N, M = p.M;
    const int nv = (N - n < 4) ? (N - n) : 4;
    long long crow = m;
    if (e.cmap) {
        const int plane = e.cOH * e.cOW;
        const int n_img = m / plane, rem = m - n_img * plane;
        const int oy = rem / e.cOW, ox = rem - oy * e.cOW;
        crow = ((long long)n_img * e.cH + (long long)oy * e.cst) * e.cW + (long long)ox * e.cst;
    }
    const float rs = e.rscale ? e.alpha * e.rscale[m] : e.alpha;
    float v0 = a[0] * rs, v1 = a[1] * rs, v2 = a[2] * rs, v3 = a[3] * rs;
    if (e.scale) {
        v0 *= e.scale[n];
        if (nv > 1) v1 *= e.scale[n + 1];
        if (nv > 2) v2 *= e.scale[n + 2];
        if (nv > 3) v3 *= e.scale[n + 3];
    }
    if (e.shift) {
        v0 += e.shift[n];
        if (nv > 1) v1 += e.shift[n + 1];
        if (nv > 2) v2 += e.shift[n + 2];
        if (nv > 3) v3 += e.shift[n + 3];
    }
    const unsigned long long didx = ((unsigned long long)bz * M + m) * N + n;
    const unsigned long long dseed = e.drop_where ? e.drop_seed + (e.drop_seed_dev ? *e.drop_seed_dev : 0ull) : 0ull;
    if (e.drop_where == 1) {
        const unsigned th = (unsigned)(e.drop_p * 4294967296.0);
        const float sc = 1.f / (1.f - e.drop_p);
        v0 = dropout_keep(dseed, didx, th) ? v0 * sc : 0.f;
        v1 = dropout_keep(dseed, didx + 1, th) ? v1 * sc : 0.f;
        v2 = dropout_keep(dseed, didx + 2, th) ? v2 * sc : 0.f;
        v3 = dropout_keep(dseed, didx + 3, th) ? v3 * sc : 0.f;
    }
    if (e.res) {
        const long long rrow = (e.res_div > 0) ? (long long)(m / e.res_div) * e.res_mod + (m % e.res_mod) : crow;
        const bf16_t* rp = (const bf16_t*)e.res + coff + rrow * e.ldr + n;
        if (nv == 4 && ((((size_t)rp) & 7) == 0)) {
            const uint2 u = *reinterpret_cast<const uint2*>(rp);
            v0 += __uint_as_float(u.x << 16); v1 += __uint_as_float(u.x & 0xffff0000u);
            v2 += __uint_as_float(u.y << 16); v3 += __uint_as_float(u.y & 0xffff0000u);
        } else {
            v0 += bf2f(rp[0]);
            if (nv > 1) v1 += bf2f(rp[1]);
            if (nv > 2) v2 += bf2f(rp[2]);
            if (nv > 3) v3 += bf2f(rp[3]);
        }
    }
    if (e.pre_out) {
        bf16_t* pp = (bf16_t*)e.pre_out + coff + crow * p.ldc + n;
        pp[0] = f2bf(v0);
        if (nv > 1) pp[1] = f2bf(v1);
        if (nv > 2) pp[2] = f2bf(v2);
        if (nv > 3) pp[3] = f2bf(v3);
    }
    if (e.act != TOIST_ACT_NONE) {
        float x0 = 0.f, x1 = 0.f, x2 = 0.f, x3 = 0.f;
        if (e.act >= TOIST_ACT_MASK_POS) {
            const bf16_t* ap = (const bf16_t*)e.aux + coff + crow * e.ldaux + n;
            if (nv == 4 && ((((size_t)ap) & 7) == 0)) {
                const uint2 u = *reinterpret_cast<const uint2*>(ap);
                x0 = __uint_as_float(u.x << 16); x1 = __uint_as_float(u.x & 0xffff0000u);
                x2 = __uint_as_float(u.y << 16); x3 = __uint_as_float(u.y & 0xffff0000u);
            } else {
                x0 = bf2f(ap[0]);
                if (nv > 1) x1 = bf2f(ap[1]);
                if (nv > 2) x2 = bf2f(ap[2]);
                if (nv > 3) x3 = bf2f(ap[3]);
            }
        }
        switch (e.act) {
            case TOIST_ACT_RELU: v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); break;
            case TOIST_ACT_GELU: v0 = gelu_f(v0); v1 = gelu_f(v1); v2 = gelu_f(v2); v3 = gelu_f(v3); break;
            case TOIST_ACT_SIGMOID:
                v0 = 1.f / (1.f + __expf(-v0)); v1 = 1.f / (1.f + __expf(-v1));
                v2 = 1.f / (1.f + __expf(-v2)); v3 = 1.f / (1.f + __expf(-v3)); break;
            case TOIST_ACT_MASK_POS:
                v0 = x0 > 0.f ? v0 : 0.f; v1 = x1 > 0.f ? v1 : 0.f; v2 = x2 > 0.f ? v2 : 0.f; v3 = x3 > 0.f ? v3 : 0.f; break;
            case TOIST_ACT_GELU_BWD:
                v0 *= gelu_grad_f(x0); v1 *= gelu_grad_f(x1); v2 *= gelu_grad_f(x2); v3 *= gelu_grad_f(x3); break;
            case TOIST_ACT_SIGMOID_BWD:
                v0 *= x0 * (1.f - x0); v1 *= x1 * (1.f - x1); v2 *= x2 * (1.f - x2); v3 *= x3 * (1.f - x3); break;
            default: break;
        }
    }
    if (e.drop_where == 2) {
        const unsigned th = (unsigned)(e.drop_p * 4294967296.0);
        const float sc = 1.f / (1.f - e.drop_p);
        v0 = dropout_keep(dseed, didx, th) ? v0 * sc : 0.f;
        v1 = dropout_keep(dseed, didx + 1, th) ? v1 * sc : 0.f;
        v2 = dropout_keep(dseed, didx + 2, th) ? v2 * sc : 0.f;
        v3 = dropout_keep(dseed, didx + 3, th) ? v3 * sc : 0.f;
    }
    if (e.out_f32) {
        float* cp = (float*)p.c + coff + crow * p.ldc + n;
        if (e.accumulate) {  // the element is owned by this thread: plain read-modify-write
            cp[0] += v0;
            if (nv > 1) cp[1] += v1;
            if (nv > 2) cp[2] += v2;
            if (nv > 3) cp[3] += v3;
        } else if (nv == 4 && ((((size_t)cp) & 15) == 0)) {
            *reinterpret_cast<float4*>(cp) = make_float4(v0, v1, v2, v3);
        } else {
            cp[0] = v0;
            if (nv > 1) cp[1] = v1;
            if (nv > 2) cp[2] = v2;
            if (nv > 3) cp[3] = v3;
        }
    } else {
        bf16_t* cp = (bf16_t*)p.c + coff + crow * p.ldc + n;
        if (nv == 4 && ((((size_t)cp) & 7) == 0)) {
            *reinterpret_cast<uint2*>(cp) = make_uint2(pack2bf(v0, v1), pack2bf(v2, v3));
        } else {
            cp[0] = f2bf(v0);
            if (nv > 1) cp[1] = f2bf(v1);
            if (nv > 2) cp[2] = f2bf(v2);
            if (nv > 3) cp[3] = f2bf(v3);
        }
    }
}

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

__device__ __forceinline__ void add8(float* acc, const u32x4_t& v) {
    acc[0] += __uint_as_float(v[0] << 16); acc[1] += __uint_as_float(v[0] & 0xffff0000u);
    acc[2] += __uint_as_float(v[1] << 16); acc[3] += __uint_as_float(v[1] & 0xffff0000u);
    acc[4] += __uint_as_float(v[2] << 16); acc[5] += __uint_as_float(v[2] & 0xffff0000u);
    acc[6] += __uint_as_float(v[3] << 16); acc[7] += __uint_as_float(v[3] & 0xffff0000u);
}

// Operand tiles go global -> LDS directly (LDS-DMA, `buffer_load_dwordx4 ... lds`): each wave
// instruction lands 64 x 16 B = 1 KiB at M0 + lane*16, so an LDS tile is "lane linear" and any
// bank-conflict swizzle is applied to WHICH global chunk a lane fetches, with the matching XOR on the
// fragment read.  The buffer descriptor spans 2 GiB from the (per-batch) base pointer; an out-of-tile
// or padding chunk uses an offset beyond it and the hardware writes zeros -- no branches, no VGPR
// staging, no ds_write pass.  Loads are counted by hand (`s_waitcnt vmcnt(N)` + raw `s_barrier`):
// hipcc neither sees the DMA nor may it drain it (cdna_hip_programming.md 5, 5.7).  Rules kept here:
// no compiler-visible VMEM access between the first DMA and the final vmcnt(0); a tile is read only
// after (own vmcnt wait) -> s_barrier; a ring slot is re-filled only after the barrier that follows
// its last read.  tools/probe/dma_probe.hip pins the DMA semantics on the hardware.
typedef __attribute__((ext_vector_type(4))) int i32x4_t;
constexpr int OOB = (int)0x80000000u;  // unsigned 2^31 >= num_records -> out of range for every dword
__device__ __forceinline__ i32x4_t make_rsrc(const void* p) {
    const unsigned long long a = (unsigned long long)p;
    i32x4_t r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(a & 0xffffffffu));
    r[1] = __builtin_amdgcn_readfirstlane((int)((a >> 32) & 0xffffu));
    r[2] = 0x7ffffff0;   // num_records (bytes): an offset at or above it reads as zero
    r[3] = 0x00020000;
    return r;
}
// one 1 KiB piece: lane l copies 16 B from (descriptor base + voff) to LDS byte address lds_dst + 16*l
__device__ __forceinline__ void dma16(unsigned lds_dst, const i32x4_t& r, int elem_off, bool valid) {
    const int voff = valid ? elem_off * 2 : OOB;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_dst), "v"(voff), "s"(r) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (N > 0) {
        static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

// ---- LDS tile layouts ------------------------------------------------------------------------------
// k-contiguous tile: [ROWS][BK], 16-byte chunk (row, kc) lives in slot kc ^ ((row / RPL) % CPR) of its row
// (CPR = BK/8 chunks per row, RPL = 16/CPR rows per 256-byte LDS line): the 16 lanes of a ds_read_b128
// group (16 consecutive rows, same kc) hit 16 distinct 16-byte slots.
template <int BK>
__device__ __forceinline__ int swz_k(int row, int kc) {
    constexpr int CPR = BK / 8, RPL = 16 / CPR;
    return kc ^ ((row / RPL) % CPR);
}
// k-major tile: [BK][ROWS], chunk (krow, rc) lives in slot rc ^ 2*h(krow) (32-byte pairs stay together
// for ds_read_b64_tr_b16), h = (krow & 3) | ((krow >> 3) & 1) << 2, limited to the pairs a row has.
template <int ROWS>
__device__ __forceinline__ int swz_m(int krow, int rc) {
    constexpr int PAIRS = ROWS / 16;
    const int h = ((krow & 3) | (((krow >> 3) & 1) << 2)) & (PAIRS - 1);
    return rc ^ (h << 1);
}

// ---- one 16-byte chunk of a staged tile ------------------------------------------------------------
struct ChunkA {            // per-thread, per-piece invariants of the A tile
    int base;              // element offset: ROWK m*lda + kc*8 ; KROW m0 + rc*8 ; conv: image n base
    int row, kc;           // tile-local (row, k-chunk) for k-contiguous tiles, (k-row, m-chunk) for k-major
    int y0, x0;            // conv gather origin
    bool ok;
};

template <int AK, int BK>
__device__ __forceinline__ void load_a(unsigned lds, const i32x4_t& rs, const ChunkA& c, const toist_operand& o, int k0, int K, int lda) {
    if (AK == TOIST_A_ROWK) { dma16(lds, rs, c.base + k0, c.ok && (k0 + c.kc * 8 < K)); return; }
    if (AK == TOIST_A_KROW) {
        const int k = k0 + c.row;
        dma16(lds, rs, c.base + k * lda, c.ok && k < K);
        return;
    }
    int tap, c0;
    if (o.SC % BK == 0) { tap = k0 / o.SC; c0 = k0 - tap * o.SC + c.kc * 8; }
    else { const int kk = k0 + c.kc * 8; tap = kk / o.SC; c0 = kk - tap * o.SC; }
    const int r = tap / o.S, s = tap - r * o.S;
    int iy, ix;
    bool in = c.ok && tap < o.R * o.S;
    if (AK == TOIST_A_CONVT) {
        const int ty = c.y0 - r * o.dil, tx = c.x0 - s * o.dil;
        in = in && (ty >= 0) && (tx >= 0);
        if (o.stride > 1) {
            in = in && ((ty % o.stride) == 0) && ((tx % o.stride) == 0);
            iy = ty / o.stride; ix = tx / o.stride;
        } else { iy = ty; ix = tx; }
    } else {
        iy = c.y0 + r * o.dil; ix = c.x0 + s * o.dil;
        in = in && (iy >= 0) && (ix >= 0);
    }
    in = in && iy < o.SH && ix < o.SW;
    dma16(lds, rs, c.base + (iy * o.SW + ix) * o.SC + c0, in);
}

struct ChunkB {
    int base;              // element offset: ROWK n*ldb + kc*8 ; KROW n0 + rc*8 ; CONVX channel offset
    int row, kc;
    int r, s;              // CONVX: tap of this chunk's columns
    bool ok;
};

template <int BKD, int BK>
__device__ __forceinline__ void load_b(unsigned lds, const i32x4_t& rs, const ChunkB& c, const toist_operand& o, int k0, int K, int ldb) {
    if (BKD == TOIST_B_ROWK) { dma16(lds, rs, c.base + k0, c.ok && (k0 + c.kc * 8 < K)); return; }
    if (BKD == TOIST_B_KROW) {
        const int k = k0 + c.row;
        int off;
        if (o.kin > 0) {
            const int tap = (o.kin % BK == 0) ? k0 / o.kin : k / o.kin;   // tile-uniform when a k-tile never straddles taps
            off = c.base + (k - tap * o.kin) * ldb + tap * (int)o.tap_stride;
        } else off = c.base + k * ldb;
        dma16(lds, rs, off, c.ok && k < K);
        return;
    }
    // CONVX: k = output pixel, columns = (tap, c)
    const int pix = k0 + c.row;
    const int plane = o.PH * o.PW;
    const int n = pix / plane;
    const int rem = pix - n * plane;
    const int py = rem / o.PW, px = rem - py * o.PW;
    const int iy = py * o.stride - o.pad + c.r * o.dil, ix = px * o.stride - o.pad + c.s * o.dil;
    const bool in = c.ok && pix < K && iy >= 0 && ix >= 0 && iy < o.SH && ix < o.SW;
    dma16(lds, rs, c.base + ((n * o.SH + iy) * o.SW + ix) * o.SC, in);
}

// MFMA operand fragment: 8 consecutive k (k = 32*ks + 8*g + j) of tile row (r0 + c16)
template <bool KM, int ROWS, int BK>
__device__ __forceinline__ bf16x8_t fragment(const bf16_t* s, int r0, int ks, int g, int c16) {
    if (KM) {
        // 16-lane group g transposes the [4 k][16 rows] blocks at k = 8g and k = 8g + 4 (ds_read_b64_tr_b16)
        const int k = ks * 32 + 8 * g + (c16 >> 2);
        const int rc = (r0 >> 3) + ((c16 & 3) >> 1);          // 16-byte chunk holding this lane's 4 rows
        const int sub = (c16 & 1) * 4;                         // element offset inside the chunk
        typedef __attribute__((address_space(3))) s16x4_t* lds_v4;
        const bf16_t* q0 = &s[k * ROWS + swz_m<ROWS>(k, rc) * 8 + sub];
        const bf16_t* q1 = &s[(k + 4) * ROWS + swz_m<ROWS>(k + 4, rc) * 8 + sub];
        const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(q0));
        const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(q1));
        union { struct { s16x4_t a, b; } h; bf16x8_t v; } u;
        u.h.a = lo; u.h.b = hi;
        return u.v;
    }
    const int row = r0 + c16;
    return *reinterpret_cast<const bf16x8_t*>(&s[row * BK + swz_k<BK>(row, ks * 4 + g) * 8]);
}

template <int BM, int BN, int BK, int AK, int BKD, int DEEP>
__global__ __launch_bounds__(256) void gemm_kernel(const toist_gemm p) {
    constexpr int WM = BM / 2, WN = BN / 2, FM = WM / 16, FN = WN / 16;
    constexpr int ACH = BM * BK / 8 / 256, BCH = BN * BK / 8 / 256;  // 1 KiB DMA pieces per wave per tile
    constexpr bool A_KM = (AK == TOIST_A_KROW);    // A staged k-major
    constexpr bool B_KM = (BKD != TOIST_B_ROWK);   // B staged k-major
    constexpr int SA_ELEMS = BM * BK, SB_ELEMS = BN * BK;
    constexpr int STAGE = SA_ELEMS + SB_ELEMS;                       // elements per ring slot
    // ring depth: measured on MI355X, occupancy beats depth -- 2 slots (5 workgroups/CU for 64x64x64) run
    // 15-30 % faster than 3 slots (3 workgroups/CU) on the K = 256..2048 hot-path shapes
    // DEEP (small grids, <= 2 workgroups per CU): nothing else hides latency, so spend the idle LDS on a 4-slot ring
    constexpr int NS = DEEP ? 4 : ((STAGE * 2 <= 8192) ? 4 : 2);
    constexpr int CNT = ACH + BCH;
    static_assert(NS >= 2 && NS <= 4, "wait ladder below covers up to 2 younger tiles");
    __shared__ __attribute__((aligned(16))) bf16_t smem[NS * STAGE];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int g = lane >> 4, c16 = lane & 15;

    const int M = p.M, N = p.N, K = p.K;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int z = blockIdx.z;
    const int bz = z / p.split_k, ksl = z - bz * p.split_k;
    const int bo = bz / p.batch_inner, bi = bz - bo * p.batch_inner;
    const toist_operand oa = p.a, ob = p.b;
    const i32x4_t rsA = make_rsrc((const bf16_t*)oa.ptr + bo * oa.bs_outer + bi * oa.bs_inner);
    const i32x4_t rsB = make_rsrc((const bf16_t*)ob.ptr + bo * ob.bs_outer + bi * ob.bs_inner);
    const long long coff = bo * p.cs_outer + bi * p.cs_inner;
    const int lda = oa.ld, ldb = ob.ld;

    const int ktiles = (K + BK - 1) / BK;
    const int kper = (ktiles + p.split_k - 1) / p.split_k;
    const int kt_beg = ksl * kper;
    const int kt_end = (kt_beg + kper < ktiles) ? kt_beg + kper : ktiles;
    if (kt_beg >= kt_end) return;

    // ---- per-thread piece invariants: piece i of wave w covers LDS chunks (i*4 + w)*64 + lane ----------
    ChunkA ca[ACH];
#pragma unroll
    for (int it = 0; it < ACH; ++it) {
        const int pch = (it * 4 + wave) * 64 + lane;
        ChunkA c;
        c.y0 = c.x0 = 0;
        if (A_KM) {
            c.row = pch / (BM / 8);                                   // k-row
            c.kc = swz_m<BM>(c.row, pch % (BM / 8));                  // m-chunk stored in this slot
            c.ok = (m0 + c.kc * 8) < M;
            c.base = m0 + c.kc * 8;
        } else {
            c.row = pch / (BK / 8);
            c.kc = swz_k<BK>(c.row, pch % (BK / 8));                  // k-chunk stored in this slot
            const int m = m0 + c.row;
            c.ok = m < M;
            if (AK == TOIST_A_ROWK) c.base = m * lda + c.kc * 8;
            else {
                const int plane = oa.PH * oa.PW;
                const int n = m / plane, rem = m - n * plane;
                const int py = rem / oa.PW, px = rem - py * oa.PW;
                c.base = n * oa.SH * oa.SW * oa.SC;
                if (AK == TOIST_A_CONVT) { c.y0 = py + oa.pad; c.x0 = px + oa.pad; }
                else { c.y0 = py * oa.stride - oa.pad; c.x0 = px * oa.stride - oa.pad; }
            }
        }
        ca[it] = c;
    }
    ChunkB cb[BCH];
#pragma unroll
    for (int it = 0; it < BCH; ++it) {
        const int pch = (it * 4 + wave) * 64 + lane;
        ChunkB c;
        c.r = c.s = 0;
        if (B_KM) {
            c.row = pch / (BN / 8);
            c.kc = swz_m<BN>(c.row, pch % (BN / 8));
            const int nn = n0 + c.kc * 8;
            c.ok = nn < N;
            if (BKD == TOIST_B_CONVX) {
                const int tap = nn / ob.SC;
                c.r = tap / ob.S; c.s = tap - c.r * ob.S;
                c.base = nn - tap * ob.SC;
            } else c.base = nn;
        } else {
            c.row = pch / (BK / 8);
            c.kc = swz_k<BK>(c.row, pch % (BK / 8));
            const int n = n0 + c.row;
            c.ok = n < N;
            c.base = n * ldb + c.kc * 8;
        }
        cb[it] = c;
    }

    f32x4_t acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    float csum[ACH][8];
    const bool want_csum = A_KM && p.a_colsum != nullptr && blockIdx.y == 0;
#pragma unroll
    for (int it = 0; it < ACH; ++it)
#pragma unroll
        for (int j = 0; j < 8; ++j) csum[it][j] = 0.f;

    const unsigned lds0 = (unsigned)(size_t)smem;  // LDS byte address of the ring
    auto issue = [&](int kt, int slot) {
        const unsigned sbase = lds0 + (unsigned)slot * (STAGE * 2) + (unsigned)wave * 1024u;
#pragma unroll
        for (int it = 0; it < ACH; ++it) load_a<AK, BK>(sbase + it * 4096u, rsA, ca[it], oa, kt * BK, K, lda);
#pragma unroll
        for (int it = 0; it < BCH; ++it) load_b<BKD, BK>(sbase + SA_ELEMS * 2 + it * 4096u, rsB, cb[it], ob, kt * BK, K, ldb);
    };

    const int ntiles = kt_end - kt_beg;
#pragma unroll
    for (int t = 0; t < NS - 1; ++t)
        if (t < ntiles) issue(kt_beg + t, t);

    int slot = 0;
    for (int t = 0; t < ntiles; ++t) {
        // tiles issued after tile t that may stay in flight: t+1 .. min(t+NS-2, ntiles-1)
        int younger = ntiles - 1 - t;
        if (younger > NS - 2) younger = NS - 2;
        if (younger <= 0) wait_vm<0>();
        else if (younger == 1) wait_vm<CNT>();
        else wait_vm<2 * CNT>();
        __builtin_amdgcn_s_barrier();   // every wave's pieces of tile t landed; everyone is done with tile t-1
        if (t + NS - 1 < ntiles) {
            int ns = slot + NS - 1;
            if (ns >= NS) ns -= NS;
            issue(kt_beg + t + NS - 1, ns);  // refills the slot tile t-1 was read from
        }
        const bf16_t* sA = smem + slot * STAGE;
        const bf16_t* sB = sA + SA_ELEMS;
        if (want_csum) {
#pragma unroll
            for (int it = 0; it < ACH; ++it) {
                const int pch = (it * 4 + wave) * 64 + lane;
                add8(csum[it], *reinterpret_cast<const u32x4_t*>(sA + pch * 8));
            }
        }
#pragma unroll
        for (int ks = 0; ks < BK / 32; ++ks) {
            bf16x8_t af[FM], bfr[FN];
#pragma unroll
            for (int i = 0; i < FM; ++i) af[i] = fragment<A_KM, BM, BK>(sA, wm * WM + i * 16, ks, g, c16);
#pragma unroll
            for (int j = 0; j < FN; ++j) bfr[j] = fragment<B_KM, BN, BK>(sB, wn * WN + j * 16, ks, g, c16);
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
        }
        if (++slot == NS) slot = 0;
    }
    wait_vm<0>();

    if (A_KM && p.a_colsum != nullptr && blockIdx.y == 0) {
        // bias gradient: column sums of the staged A tiles, reduced over the k rows held by other threads
        float* red = reinterpret_cast<float*>(smem);  // main loop is done: LDS is free
        __syncthreads();
#pragma unroll
        for (int it = 0; it < ACH; ++it)
#pragma unroll
            for (int j = 0; j < 8; ++j) red[ca[it].row * BM + ca[it].kc * 8 + j] = csum[it][j];
        __syncthreads();
        for (int col = tid; col < BM; col += 256) {
            float t = 0.f;
            for (int r = 0; r < BK; ++r) t += red[r * BM + col];
            if (m0 + col < M) atomicAdd(p.a_colsum + m0 + col, t);
        }
    }

    // ---- epilogue: lane owns output row m (c16) and 4 consecutive columns n (4*g .. 4*g+3) ----
    if (p.split_k > 1) {
        // k-slice partial: raw f32 tile into the workspace; splitk_reduce_kernel applies the epilogue
        float* ws = p.workspace + (size_t)ksl * M * N;
        static_for<FM * FN>([&](auto idx) {
            constexpr int i = decltype(idx)::value / FN, j = decltype(idx)::value % FN;
            const int m = m0 + wm * WM + i * 16 + c16;
            const int n = n0 + wn * WN + j * 16 + g * 4;
            if (m < M && n < N) {
                float* cp = ws + (size_t)m * N + n;
                const f32x4_t a = acc[i][j];
                if (N - n >= 4 && ((((size_t)cp) & 15) == 0)) *reinterpret_cast<float4*>(cp) = make_float4(a[0], a[1], a[2], a[3]);
                else {
                    cp[0] = a[0];
                    if (N - n > 1) cp[1] = a[1];
                    if (N - n > 2) cp[2] = a[2];
                    if (N - n > 3) cp[3] = a[3];
                }
            }
        });
        return;
    }
    static_for<FM * FN>([&](auto idx) {
        constexpr int i = decltype(idx)::value / FN, j = decltype(idx)::value % FN;
        const int m = m0 + wm * WM + i * 16 + c16;
        const int n = n0 + wn * WN + j * 16 + g * 4;
        if (m < M && n < N) epilogue_frag(p, acc[i][j], m, n, bz, coff);
    });
}

// C[m][n] (+)= alpha * rscale[m] * sum_s ws[s][m][n]   (second half of a split-K GEMM)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, int splits, int M, int N, float alpha,
                                                             const float* __restrict__ rscale, int accumulate, float* __restrict__ c,
                                                             int ldc) {
    const long long total4 = ((long long)M * N) >> 2;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long long)gridDim.x * 256) {
        float4 s = reinterpret_cast<const float4*>(ws)[i];
        for (int k = 1; k < splits; ++k) {
            const float4 t = reinterpret_cast<const float4*>(ws + (size_t)k * M * N)[i];
            s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
        }
        const long long e = i << 2;
        const int m = (int)(e / N), n = (int)(e - (long long)m * N);
        const float f = rscale ? alpha * rscale[m] : alpha;
        float* cp = c + (long long)m * ldc + n;
        if (accumulate) { cp[0] += s.x * f; cp[1] += s.y * f; cp[2] += s.z * f; cp[3] += s.w * f; }
        else { cp[0] = s.x * f; cp[1] = s.y * f; cp[2] = s.z * f; cp[3] = s.w * f; }
    }
}

template <int BM, int BN, int BK, int AK, int BKD>
static void launch_variant(const toist_gemm& d, hipStream_t st) {
    dim3 grid((d.M + BM - 1) / BM, (d.N + BN - 1) / BN, d.batch * d.split_k);
    const long long wgs = (long long)grid.x * grid.y * grid.z;
    if (BM == 64 && BN == 64 && BK == 64 && wgs <= 512)
        hipLaunchKernelGGL((gemm_kernel<BM, BN, BK, AK, BKD, (BM == 64 && BN == 64 && BK == 64) ? 1 : 0>), grid, dim3(256), 0, st, d);
    else
        hipLaunchKernelGGL((gemm_kernel<BM, BN, BK, AK, BKD, 0>), grid, dim3(256), 0, st, d);
}

template <int BM, int BN, int BK>
static int launch_tile(const toist_gemm& d, hipStream_t st) {
    const int ak = d.a_kind, bk = d.b_kind;
    if (ak == TOIST_A_ROWK && bk == TOIST_B_ROWK) launch_variant<BM, BN, BK, TOIST_A_ROWK, TOIST_B_ROWK>(d, st);
    else if (ak == TOIST_A_ROWK && bk == TOIST_B_KROW) launch_variant<BM, BN, BK, TOIST_A_ROWK, TOIST_B_KROW>(d, st);
    else if (ak == TOIST_A_CONV && bk == TOIST_B_ROWK) launch_variant<BM, BN, BK, TOIST_A_CONV, TOIST_B_ROWK>(d, st);
    else if (ak == TOIST_A_CONVT && bk == TOIST_B_KROW) launch_variant<BM, BN, BK, TOIST_A_CONVT, TOIST_B_KROW>(d, st);
    else if (ak == TOIST_A_KROW && bk == TOIST_B_KROW) launch_variant<BM, BN, BK, TOIST_A_KROW, TOIST_B_KROW>(d, st);
    else if (ak == TOIST_A_KROW && bk == TOIST_B_CONVX) launch_variant<BM, BN, BK, TOIST_A_KROW, TOIST_B_CONVX>(d, st);
    else {
        set_last_error("toist_gemm_bf16: unsupported operand kinds a=%d b=%d", ak, bk);
        return TOIST_EINVAL;
    }
    return TOIST_OK;
}

static bool aligned16(const void* p) { return (((size_t)p) & 15) == 0; }

}  // namespace toist

extern "C" int toist_gemm_bf16(const toist_gemm* desc, void* stream) {
    using namespace toist;
    TOIST_REQUIRE(desc != nullptr, "toist_gemm_bf16: null descriptor");
    toist_gemm d = *desc;
    TOIST_REQUIRE(d.M > 0 && d.N > 0 && d.K > 0, "toist_gemm_bf16: bad shape M=%d N=%d K=%d", d.M, d.N, d.K);
    if (d.batch <= 0) d.batch = 1;
    if (d.batch_inner <= 0) d.batch_inner = 1;
    if (d.split_k <= 0) d.split_k = 1;
    TOIST_REQUIRE(d.a.ptr && d.b.ptr && d.c, "toist_gemm_bf16: null operand");
    TOIST_REQUIRE(aligned16(d.a.ptr) && aligned16(d.b.ptr), "toist_gemm_bf16: A/B must be 16-byte aligned");
    const toist_operand* ops[2] = {&d.a, &d.b};
    const int kinds[2] = {d.a_kind, d.b_kind == TOIST_B_CONVX ? TOIST_A_CONV : (d.b_kind == TOIST_B_KROW ? TOIST_A_KROW : TOIST_A_ROWK)};
    for (int i = 0; i < 2; ++i) {
        const toist_operand& o = *ops[i];
        TOIST_REQUIRE((o.bs_outer % 8) == 0 && (o.bs_inner % 8) == 0, "toist_gemm_bf16: batch strides must be multiples of 8 elements");
        if (kinds[i] == TOIST_A_CONV || kinds[i] == TOIST_A_CONVT) {
            TOIST_REQUIRE(o.SC > 0 && (o.SC % 8) == 0, "toist_gemm_bf16: source channels must be a multiple of 8 (got %d)", o.SC);
            TOIST_REQUIRE(o.R > 0 && o.S > 0 && o.stride > 0 && o.dil > 0 && o.PH > 0 && o.PW > 0 && o.SH > 0 && o.SW > 0,
                          "toist_gemm_bf16: bad conv geometry");
        } else {
            TOIST_REQUIRE(o.ld > 0 && (o.ld % 8) == 0, "toist_gemm_bf16: leading dimension must be a multiple of 8 (got %d)", o.ld);
        }
    }
    if (d.b_kind == TOIST_B_CONVX) TOIST_REQUIRE((d.N % 8) == 0, "toist_gemm_bf16: CONVX needs N %% 8 == 0");
    // k-major operands are read in 8-row chunks: rows beyond M/N inside the last chunk are read
    // (and discarded), so ld must cover the rounded-up extent.
    if (d.a_kind == TOIST_A_KROW) TOIST_REQUIRE(d.a.ld >= ((d.M + 7) & ~7), "toist_gemm_bf16: A_KROW needs lda >= roundup8(M)");
    if (d.b_kind == TOIST_B_KROW) TOIST_REQUIRE(d.b.ld >= ((d.N + 7) & ~7), "toist_gemm_bf16: B_KROW needs ldb >= roundup8(N)");
    if (d.split_k > 1 || d.epi.accumulate)
        TOIST_REQUIRE(d.epi.out_f32, "toist_gemm_bf16: split_k/accumulate needs an f32 output");
    if (d.a_colsum) TOIST_REQUIRE(d.a_kind == TOIST_A_KROW, "toist_gemm_bf16: a_colsum needs a k-major A operand");
    if (d.split_k > 1) {
        TOIST_REQUIRE(!d.epi.scale && !d.epi.shift && !d.epi.res && d.epi.act == TOIST_ACT_NONE && !d.epi.pre_out && d.epi.drop_where == 0 &&
                          !d.epi.cmap && d.batch == 1,
                      "toist_gemm_bf16: split_k only supports alpha/rscale/accumulate epilogues on a single batch");
        TOIST_REQUIRE(d.workspace != nullptr && (d.N % 4) == 0 && (d.ldc % 4) == 0, "toist_gemm_bf16: split_k needs a workspace and N, ldc %% 4 == 0");
    }
    if (d.epi.act >= TOIST_ACT_MASK_POS) TOIST_REQUIRE(d.epi.aux != nullptr, "toist_gemm_bf16: activation %d needs aux", d.epi.act);
    if (d.epi.drop_where) TOIST_REQUIRE(d.epi.drop_p >= 0.f && d.epi.drop_p < 1.f, "toist_gemm_bf16: bad dropout p");

    int tile = d.tile;
    if (tile == 0) {
        // measured on MI355X (tools/sweep_gemm.py): 128x128x64 only pays once >= ~4 tiles per CU exist and K
        // is deep; below that 64x64x64 tiles keep more workgroups (and DMA) in flight.
        const long long t128 = (long long)((d.M + 127) / 128) * ((d.N + 127) / 128) * d.batch * d.split_k;
        if (t128 >= 1024 && d.K >= 1024) tile = 129;
        else tile = (d.K > 64) ? 65 : 64;
    }
    {
        // drop k-slices that would own no k-tile
        const int bk0 = (tile == 64 || tile == 128) ? 32 : 64;
        const int ktiles = (d.K + bk0 - 1) / bk0;
        if (d.split_k > ktiles) d.split_k = ktiles;
        const int kper = (ktiles + d.split_k - 1) / d.split_k;
        d.split_k = (ktiles + kper - 1) / kper;
    }
    // tile codes: 64 = 64x64x32, 65 = 64x64x64, 128 = 128x128x32, 129 = 128x128x64, 130 = 128x64x64
    const int bkt = (tile == 64 || tile == 128) ? 32 : 64;
    if (d.b_kind == TOIST_B_KROW && d.b.kin > 0) TOIST_REQUIRE((d.b.kin % 8) == 0, "toist_gemm_bf16: kin %% 8 != 0");
    (void)bkt;
    int rc;
    hipStream_t st = (hipStream_t)stream;
    switch (tile) {
        case 64: rc = launch_tile<64, 64, 32>(d, st); break;
        case 65: rc = launch_tile<64, 64, 64>(d, st); break;
        case 128: rc = launch_tile<128, 128, 32>(d, st); break;
        case 129: rc = launch_tile<128, 128, 64>(d, st); break;
        case 130: rc = launch_tile<128, 64, 64>(d, st); break;
        default: set_last_error("toist_gemm_bf16: bad tile code %d", tile); return TOIST_EINVAL;
    }
    if (rc != TOIST_OK) return rc;
    rc = check_launch("toist_gemm_bf16");
    if (rc != TOIST_OK || d.split_k <= 1) return rc;
    const long long total4 = ((long long)d.M * d.N) / 4;
    int grid = (int)((total4 + 255) / 256 > 2048 ? 2048 : (total4 + 255) / 256);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(grid), dim3(256), 0, st, (const float*)d.workspace, d.split_k, d.M, d.N, d.epi.alpha,
                       d.epi.rscale, d.epi.accumulate, (float*)d.c, d.ldc);
    return check_launch("toist_gemm_bf16(splitk reduce)");
}
