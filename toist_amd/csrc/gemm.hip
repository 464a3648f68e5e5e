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
#include <cstdlib>
#include <type_traits>

#include "common.h"

namespace toist {

// BK (template parameter) = k extent of one staged tile; k-contiguous LDS tiles use a row pitch of
// BK + 8 elements (80 / 144 B) so the 16-lane groups of a ds_read_b128 spread over the banks.

__device__ __forceinline__ float gelu_f(float v) { return 0.5f * v * (1.f + erff(v * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad_f(float a) {
    return 0.5f * (1.f + erff(a * 0.70710678118654752f)) + a * 0.3989422804014327f * __expf(-0.5f * a * a);
}

template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (N > 0) {
        static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

// One 1x8 output chunk (row m, columns n .. n+7; the accumulators reach it through an LDS transpose so that a
// wavefront touches whole 128-byte row segments):
//   v = acc*alpha*rscale[m] -> *scale[n] + shift[n] -> dropout -> + res -> (pre_out) -> act -> dropout -> store.
// Written with compile-time element indices only (no break/continue) so everything stays in registers.
struct EpiRow {            // addresses of one chunk, resolved once
    long long crow;        // output row (after the optional scatter map)
    long long rrow;        // residual row
    int m, n, nv;          // nv = valid columns (<= 8)
};

__device__ __forceinline__ EpiRow epi_row(const toist_gemm& p, const int m, const int n) {
    const toist_epilogue& e = p.epi;
    EpiRow r;
    r.m = m; r.n = n;
    r.nv = (p.N - n < 8) ? (p.N - n) : 8;
    r.crow = m;
    if (e.cmap) {
        const int plane = e.cOH * e.cOW;
        const int n_img = m / plane, rem = m - n_img * plane;
        const int oy = rem / e.cOW, ox = rem - oy * e.cOW;
        r.crow = ((long long)n_img * e.cH + (long long)oy * e.cst) * e.cW + (long long)ox * e.cst;
    }
    r.rrow = (e.res_div > 0) ? (long long)(m / e.res_div) * e.res_mod + (m % e.res_mod) : r.crow;
    return r;
}

__device__ __forceinline__ void unpack8(const uint4 u, float* x) {
    x[0] = __uint_as_float(u.x << 16); x[1] = __uint_as_float(u.x & 0xffff0000u);
    x[2] = __uint_as_float(u.y << 16); x[3] = __uint_as_float(u.y & 0xffff0000u);
    x[4] = __uint_as_float(u.z << 16); x[5] = __uint_as_float(u.z & 0xffff0000u);
    x[6] = __uint_as_float(u.w << 16); x[7] = __uint_as_float(u.w & 0xffff0000u);
}

// 8 bf16 of a row; whole 16-byte load when the chunk is complete and aligned
__device__ __forceinline__ void load_row8(const bf16_t* rp, const int nv, float* x) {
    if (nv == 8 && ((((size_t)rp) & 15) == 0)) {
        unpack8(*reinterpret_cast<const uint4*>(rp), x);
    } else {
        static_for<8>([&](auto jj) {
            constexpr int j = decltype(jj)::value;
            x[j] = (j < nv) ? bf2f(rp[j]) : 0.f;
        });
    }
}

#ifndef EPI_NT_STORE
#define EPI_NT_STORE 0
#endif
__device__ __forceinline__ void store_row8_bf16(bf16_t* cp, const int nv, const float* v) {
    if (nv == 8 && ((((size_t)cp) & 15) == 0)) {
#if EPI_NT_STORE
        typedef __attribute__((ext_vector_type(4))) unsigned int u4;
        const u4 val = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7])};
        __builtin_nontemporal_store(val, reinterpret_cast<u4*>(cp));
#else
        *reinterpret_cast<uint4*>(cp) = make_uint4(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7]));
#endif
    } else {
        static_for<8>([&](auto jj) {
            constexpr int j = decltype(jj)::value;
            if (j < nv) cp[j] = f2bf(v[j]);
        });
    }
}

// res / aux operands of a chunk are requested (raw 16-byte loads, nothing depends on them yet) ahead of the LDS
// transpose so their latency overlaps it; chunks that are incomplete or misaligned are loaded element-wise at use.
struct EpiPre {
    uint4 res, aux;
    bool res_vec, aux_vec;
};

// per-column epilogue vectors (scale, shift) of a thread's 8 columns: the column does not change from band to band,
// so they are read once per tile, as two 16-byte loads when the chunk is complete (8 scalar loads per chunk used to
// cost a quarter of the 1x1-conv kernels)
struct EpiCols {
    float scale[8], shift[8];
};

__device__ __forceinline__ void load_cols8(const float* src, const int nv, float* x, const float fill) {
    if (nv == 8 && ((((size_t)src) & 15) == 0)) {
        const float4 a = reinterpret_cast<const float4*>(src)[0], b = reinterpret_cast<const float4*>(src)[1];
        x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w; x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w;
    } else {
        static_for<8>([&](auto jj) {
            constexpr int j = decltype(jj)::value;
            x[j] = (j < nv) ? src[j] : fill;
        });
    }
}

__device__ __forceinline__ void epilogue_cols(const toist_gemm& p, const int n, EpiCols& c, const long long voff) {
    const toist_epilogue& e = p.epi;
    const int nv = (p.N - n < 8) ? (p.N - n) : 8;
    if (nv <= 0) return;
    if (e.scale) load_cols8(e.scale + voff + n, nv, c.scale, 1.f);
    if (e.shift) load_cols8(e.shift + voff + n, nv, c.shift, 0.f);
}

__device__ __forceinline__ bool row8_vec(const bf16_t* rp, const int nv) { return nv == 8 && ((((size_t)rp) & 15) == 0); }

__device__ __forceinline__ void epilogue_fetch(const toist_gemm& p, const EpiRow& r, const long long coff, EpiPre& pre) {
    const toist_epilogue& e = p.epi;
    pre.res_vec = pre.aux_vec = false;
    if (e.res) {
        const bf16_t* rp = (const bf16_t*)e.res + coff + r.rrow * e.ldr + r.n;
        pre.res_vec = row8_vec(rp, r.nv);
        if (pre.res_vec) pre.res = *reinterpret_cast<const uint4*>(rp);
    }
    if (e.act >= TOIST_ACT_MASK_POS) {
        const bf16_t* ap = (const bf16_t*)e.aux + coff + r.crow * e.ldaux + r.n;
        pre.aux_vec = row8_vec(ap, r.nv);
        if (pre.aux_vec) pre.aux = *reinterpret_cast<const uint4*>(ap);
    }
}

__device__ __forceinline__ void epilogue_operands(const toist_gemm& p, const EpiRow& r, const long long coff, const EpiPre& pre, float* xres,
                                                  float* xaux) {
    const toist_epilogue& e = p.epi;
    if (e.res) {
        if (pre.res_vec) unpack8(pre.res, xres);
        else load_row8((const bf16_t*)e.res + coff + r.rrow * e.ldr + r.n, r.nv, xres);
    }
    if (e.act >= TOIST_ACT_MASK_POS) {
        if (pre.aux_vec) unpack8(pre.aux, xaux);
        else load_row8((const bf16_t*)e.aux + coff + r.crow * e.ldaux + r.n, r.nv, xaux);
    }
}

__device__ __forceinline__ void epilogue_row8(const toist_gemm& p, float* v, const EpiRow& r, const int bz, const long long coff,
                                              const float* xres, const float* xaux, const EpiCols& cols) {
    const toist_epilogue& e = p.epi;
    const int N = p.N, M = p.M, m = r.m, n = r.n, nv = r.nv;
    const float* rsc = e.rscale;
    if (rsc && p.group) rsc += p.group[bz].rscale_off;
    const float rs = rsc ? e.alpha * rsc[m] : e.alpha;
    static_for<8>([&](auto jj) { v[decltype(jj)::value] *= rs; });
    if (e.scale) static_for<8>([&](auto jj) { constexpr int j = decltype(jj)::value; v[j] *= cols.scale[j]; });
    if (e.shift) static_for<8>([&](auto jj) { constexpr int j = decltype(jj)::value; v[j] += cols.shift[j]; });
    const unsigned long long didx = ((unsigned long long)bz * M + m) * N + n;
    const unsigned long long dseed = e.drop_where ? e.drop_seed + (e.drop_seed_dev ? *e.drop_seed_dev : 0ull) : 0ull;
    if (e.drop_where == 1) {
        const unsigned th = (unsigned)(e.drop_p * 4294967296.0);
        const float sc = 1.f / (1.f - e.drop_p);
        static_for<8>([&](auto jj) { constexpr int j = decltype(jj)::value; v[j] = dropout_keep(dseed, didx + j, th) ? v[j] * sc : 0.f; });
    }
    if (e.res) static_for<8>([&](auto jj) { constexpr int j = decltype(jj)::value; v[j] += xres[j]; });
    if (e.pre_out) store_row8_bf16((bf16_t*)e.pre_out + coff + r.crow * p.ldc + n, nv, v);
    switch (e.act) {
        case TOIST_ACT_RELU: static_for<8>([&](auto jj) { constexpr int j = decltype(jj)::value; v[j] = fmaxf(v[j], 0.f); }); break;
        case TOIST_ACT_GELU: static_for<8>([&](auto jj) { constexpr int j = decltype(jj)::value; v[j] = gelu_f(v[j]); }); break;
        case TOIST_ACT_SIGMOID: static_for<8>([&](auto jj) { constexpr int j = decltype(jj)::value; v[j] = 1.f / (1.f + __expf(-v[j])); }); break;
        case TOIST_ACT_MASK_POS: static_for<8>([&](auto jj) { constexpr int j = decltype(jj)::value; v[j] = xaux[j] > 0.f ? v[j] : 0.f; }); break;
        case TOIST_ACT_GELU_BWD: static_for<8>([&](auto jj) { constexpr int j = decltype(jj)::value; v[j] *= gelu_grad_f(xaux[j]); }); break;
        case TOIST_ACT_SIGMOID_BWD: static_for<8>([&](auto jj) { constexpr int j = decltype(jj)::value; v[j] *= xaux[j] * (1.f - xaux[j]); }); break;
        default: break;
    }
    if (e.drop_where == 2) {
        const unsigned th = (unsigned)(e.drop_p * 4294967296.0);
        const float sc = 1.f / (1.f - e.drop_p);
        static_for<8>([&](auto jj) { constexpr int j = decltype(jj)::value; v[j] = dropout_keep(dseed, didx + j, th) ? v[j] * sc : 0.f; });
    }
    if (e.out_f32) {
        float* cp = (float*)p.c + coff + r.crow * p.ldc + n;
        if (e.accumulate) {  // the element is owned by this thread: plain read-modify-write
            static_for<8>([&](auto jj) { constexpr int j = decltype(jj)::value; if (j < nv) cp[j] += v[j]; });
        } else if (nv == 8 && ((((size_t)cp) & 15) == 0)) {
            reinterpret_cast<float4*>(cp)[0] = make_float4(v[0], v[1], v[2], v[3]);
            reinterpret_cast<float4*>(cp)[1] = make_float4(v[4], v[5], v[6], v[7]);
        } else {
            static_for<8>([&](auto jj) { constexpr int j = decltype(jj)::value; if (j < nv) cp[j] = v[j]; });
        }
    } else {
        store_row8_bf16((bf16_t*)p.c + coff + r.crow * p.ldc + n, nv, v);
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
// the same with the tile-dependent part of the source offset in the instruction's SGPR offset: the lane offset (bytes, or OOB) is a
// loop invariant, no vector arithmetic per piece
__device__ __forceinline__ void dma16s(unsigned lds_dst, const i32x4_t& r, int voff_bytes, int soff_bytes) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_dst), "v"(voff_bytes), "s"(r), "s"(soff_bytes) : "memory");
}
// two pieces (LDS destinations 4 KiB apart: pieces it = 0, 1 of a wave) with one M0 save / restore
__device__ __forceinline__ void dma16s2(unsigned lds_dst, const i32x4_t& r, int voff0, int voff1, int soff_bytes) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %4, %5 offen lds\n\t"
                 "s_mov_b32 m0, %6\n\ts_nop 0\n\tbuffer_load_dwordx4 %3, %4, %5 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_dst), "v"(voff0), "v"(voff1), "s"(r), "s"(soff_bytes), "s"(lds_dst + 4096u) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// shader clock (experiments: per-phase cycle counters behind flags bit 11)
__device__ __forceinline__ unsigned long long cyc_now() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
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
// for ds_read_b64_tr_b16).  One ds_read_b64_tr_b16 is served per 32-lane half: the half of lane groups g = 0, 1 reads k rows
// {k0 .. k0 + 3} u {k0 + 8 .. k0 + 11}, 32 bytes each = 256 bytes, conflict-free iff the eight pieces fall on eight different 32-byte bank
// slots.  128-row tiles (256-byte rows: every row starts on bank 0) need three hash bits: h = (krow & 3) | ((krow >> 3) & 1) << 2.
// 64-row tiles (128-byte rows) only have four pairs = two hash bits, but the row PARITY already moves a row by half the banks, so the two
// bits go to krow bit 1 and krow bit 3: h = ((krow >> 1) & 1) | ((krow >> 3) & 1) << 1.  (Round 4 masked the 128-row hash to two bits = krow & 3:
// rows k and k + 8 collided -- 32-47 % of the LDS-active cycles of every k-major 64 x 64 kernel were bank conflicts,
// profiles/r04_pmc_lds_conflicts.txt.)
template <int ROWS>
__device__ __forceinline__ int swz_m(int krow, int rc) {
    constexpr int PAIRS = ROWS / 16;
    const int h = PAIRS == 4 ? (((krow >> 1) & 1) | (((krow >> 3) & 1) << 1)) : (((krow & 3) | (((krow >> 3) & 1) << 2)) & (PAIRS - 1));
    return rc ^ (h << 1);
}

// ---- one 16-byte chunk of a staged tile ------------------------------------------------------------
struct ChunkA {            // per-thread, per-piece invariants of the A tile
    int base;              // element offset: ROWK m*lda + kc*8 ; KROW m0 + rc*8 ; conv: image n base
    int row, kc;           // tile-local (row, k-chunk) for k-contiguous tiles, (k-row, m-chunk) for k-major
    int y0, x0;            // conv gather origin
    bool ok;
};

// Source walkers.  The k-loop visits k-tiles in order, so everything that depends on k is kept as running state and
// advanced once per tile instead of being re-derived (integer divisions by runtime extents, tap decomposition, bounds)
// for every 1 KiB piece: the gather arithmetic used to be 120-280 VALU/SALU instructions per k-tile against 8 MFMAs
// and bounded every convolution kernel by instruction issue.
//
// Uniform position inside a two-level reduction index k = tap * span + inner (conv gathers: span = source channels;
// k-major KRSC weights: span = kin).  `fast` = a k-tile never straddles two taps (span % BK == 0).
struct TapPos {
    int tap, inner, r, s;
    __device__ __forceinline__ void init(int k0, int span, int S) {
        tap = k0 / span; inner = k0 - tap * span; r = tap / S; s = tap - r * S;
    }
    // returns true when the tap changed
    __device__ __forceinline__ bool advance(int step, int span, int S) {
        inner += step;
        if (inner < span) return false;
        inner -= span; ++tap;
        if (++s == S) { s = 0; ++r; }
        return true;
    }
};

// per-piece, per-tap state of a conv gather: element offset of the tap's pixel (+ this lane's k-chunk) and validity
template <int AK>
__device__ __forceinline__ void conv_tap(const ChunkA& c, const toist_operand& o, const TapPos& tp, int& off, bool& in) {
    int iy, ix;
    in = c.ok && tp.tap < o.R * o.S;
    if (AK == TOIST_A_CONVT) {
        const int ty = c.y0 - tp.r * o.dil, tx = c.x0 - tp.s * o.dil;
        in = in && (ty >= 0) && (tx >= 0);
        if (o.stride == 2) { in = in && !((ty | tx) & 1); iy = ty >> 1; ix = tx >> 1; }
        else if (o.stride > 1) { in = in && ((ty % o.stride) == 0) && ((tx % o.stride) == 0); iy = ty / o.stride; ix = tx / o.stride; }
        else { iy = ty; ix = tx; }
    } else {
        iy = c.y0 + tp.r * o.dil; ix = c.x0 + tp.s * o.dil;
        in = in && (iy >= 0) && (ix >= 0);
    }
    in = in && iy < o.SH && ix < o.SW;
    off = c.base + (iy * o.SW + ix) * o.SC + c.kc * 8;
}

// slow path of the conv gathers (source channels not a multiple of BK, e.g. the 8-channel stem): per-chunk taps
template <int AK, int BK>
__device__ __forceinline__ void load_a_slow(unsigned lds, const i32x4_t& rs, const ChunkA& c, const toist_operand& o, int k0) {
    const int kk = k0 + c.kc * 8;
    TapPos tp;
    tp.tap = kk / o.SC; tp.inner = kk - tp.tap * o.SC; tp.r = tp.tap / o.S; tp.s = tp.tap - tp.r * o.S;
    int off; bool in;
    conv_tap<AK>(c, o, tp, off, in);
    dma16(lds, rs, off - c.kc * 8 + tp.inner, in);
}

struct ChunkB {
    int base;              // element offset: ROWK n*ldb + kc*8 ; KROW n0 + rc*8 ; CONVX channel offset
    int row, kc;
    int r, s;              // CONVX: tap of this chunk's columns
    bool ok;
};

// CONVX columns: pixel walk of k = output pixel index (wgrad: the reduction runs over pixels)
struct PixPos {
    int n, py, px;
    __device__ __forceinline__ void init(int pix, int PH, int PW) {
        const int plane = PH * PW;
        n = pix / plane;
        const int rem = pix - n * plane;
        py = rem / PW; px = rem - py * PW;
    }
    __device__ __forceinline__ void advance(int dq, int dr, int PH, int PW) {   // += dq rows + dr pixels
        px += dr; py += dq;
        if (px >= PW) { px -= PW; ++py; }
        while (py >= PH) { py -= PH; ++n; }
    }
};

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

// fragment<true, ...> with the rows of a 32-row group in the order panel2_kernel wants (row q of fragment j = row 8 (q >> 2) + 4 j + (q & 3) of the
// group starting at r0, a multiple of 8): lane quad q & 3 of a 16-lane group transposes the four rows at element 4 j of the group's chunk (q & 3).
template <int ROWS, int BK>
__device__ __forceinline__ bf16x8_t fragment_perm8(const bf16_t* s, int r0, int j, int ks, int g, int c16) {
    const int k = ks * 32 + 8 * g + (c16 >> 2);
    const int rc = (r0 >> 3) + (c16 & 3);
    const int sub = j * 4;
    typedef __attribute__((address_space(3))) s16x4_t* lds_v4;
    const bf16_t* q0 = &s[k * ROWS + swz_m<ROWS>(k, rc) * 8 + sub];
    const bf16_t* q1 = &s[(k + 4) * ROWS + swz_m<ROWS>(k + 4, rc) * 8 + sub];
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(q0));
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(q1));
    union { struct { s16x4_t a, b; } h; bf16x8_t v; } u;
    u.h.a = lo; u.h.b = hi;
    return u.v;
}

// logical tile index -> (tile_m, tile_n): panels of 8 M tiles, M fastest inside a panel, then N, then the next panel
__device__ __forceinline__ void tile_order(int L, int nt_m, int nt_n, int& tile_m, int& tile_n) {
    constexpr int G = 8;
    const int per = G * nt_n;
    const int panel = L / per, rem = L - panel * per;
    const int rows = (nt_m - panel * G < G) ? nt_m - panel * G : G;   // the last panel may be short
    tile_n = rem / rows;
    tile_m = panel * G + (rem - tile_n * rows);
}

// ---- epilogue of one workgroup tile -----------------------------------------------------------------------
template <int BN, int FM>
struct EpiTile {            // rows and pre-requested residual / aux chunks of one thread for every band of a tile
    static constexpr int CPR = BN / 8;                      // 8-column chunks per band row
    static constexpr int CH = (32 * CPR + 255) / 256;       // chunks per thread per band (1 for BN <= 64, 2 for BN = 128)
    static_assert(32 * CPR % 256 == 0 || 32 * CPR < 256, "band chunks must divide over 256 threads");
    EpiRow rows[FM][CH];
    EpiPre pre[FM][CH];
};

// first half: resolve this thread's rows of tile (m0, n0) and request the residual / aux operands of EVERY band up front (one
// exposed load latency per tile, not one per band; the panel kernel issues it a whole tile ahead)
template <int BN, int WM, int FM>
__device__ __forceinline__ void epilogue_request(const toist_gemm& p, const int m0, const int n0, const long long coff, const bool partial,
                                                 EpiTile<BN, FM>& t) {
    constexpr int CPR = EpiTile<BN, FM>::CPR, CH = EpiTile<BN, FM>::CH;
    const int tid = threadIdx.x, M = p.M;
    static_for<FM>([&](auto ii) {
        constexpr int i = decltype(ii)::value;
#pragma unroll
        for (int q = 0; q < CH; ++q) {
            const int c = tid + 256 * q;
            const int br = c / CPR, c8 = c - br * CPR;
            const int m = (c < 32 * CPR) ? m0 + (br >> 4) * WM + i * 16 + (br & 15) : M;   // narrow tiles: the upper threads idle
            t.rows[i][q] = epi_row(p, m < M ? m : 0, n0 + c8 * 8);
            t.rows[i][q].m = m;
            if (!partial && m < M && t.rows[i][q].nv > 0) epilogue_fetch(p, t.rows[i][q], coff, t.pre[i][q]);
        }
    });
}

// second half: the accumulators go through LDS band by band and every thread finishes 8 consecutive columns of one row
// Barrier between the LDS writes and reads of the epilogue bands.  NOT __syncthreads(): that is a workgroup-scope fence, and hipcc
// drains vmcnt in front of it -- every band then waited for the previous band's GLOBAL stores to be acknowledged (a 1-2 us round
// trip under load, four times per 64x64 tile), which is what kept the store phase of the 1x1-convolution GEMMs at 2.2 TB/s while the
// same tile pattern written without barriers runs at 4.3 (tools/probe/store_pattern.hip).  Only the LDS traffic has to be ordered.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int BN, int WM, int WN, int FM, int FN>
__device__ __forceinline__ void epilogue_finish(const toist_gemm& p, f32x4_t (&acc)[FM][FN], float* band, EpiTile<BN, FM>& t,
                                                const EpiCols (&cols)[EpiTile<BN, FM>::CH], const int bz, const long long coff, const int ksl,
                                                const bool partial) {
    constexpr int CPR = EpiTile<BN, FM>::CPR, CH = EpiTile<BN, FM>::CH;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, g = lane >> 4, c16 = lane & 15;
    const int M = p.M, N = p.N;
    constexpr int LDT = BN + 4;                      // f32 band pitch: +4 keeps the float4 writes conflict-free
    float* ws = partial ? p.workspace + ((size_t)bz * p.split_k + ksl) * M * N : nullptr;   // [problem][k-slice][M][N]
    static_for<FM>([&](auto ii) {
        constexpr int i = decltype(ii)::value;
        EpiRow (&rows)[CH] = t.rows[i];
        lds_barrier();                               // previous band (or the last k-tile / the colsum scratch) is consumed
        static_for<FN>([&](auto jj) {
            constexpr int j = decltype(jj)::value;
            *reinterpret_cast<f32x4_t*>(band + (wm * 16 + c16) * LDT + wn * WN + j * 16 + g * 4) = acc[i][j];
        });
        lds_barrier();
#pragma unroll
        for (int q = 0; q < CH; ++q) {
            const int c = tid + 256 * q;
            const int br = c / CPR, c8 = c - br * CPR;
            const EpiRow& r = rows[q];
            if (r.m < M && r.nv > 0) {
                float v[8];
                const f32x4_t lo = *reinterpret_cast<const f32x4_t*>(band + br * LDT + c8 * 8);
                const f32x4_t hi = *reinterpret_cast<const f32x4_t*>(band + br * LDT + c8 * 8 + 4);
                v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3]; v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
                if (partial) {
                    float* cp = ws + (size_t)r.m * N + r.n;
                    if (r.nv == 8 && ((((size_t)cp) & 15) == 0)) {
                        reinterpret_cast<float4*>(cp)[0] = make_float4(v[0], v[1], v[2], v[3]);
                        reinterpret_cast<float4*>(cp)[1] = make_float4(v[4], v[5], v[6], v[7]);
                    } else {
                        static_for<8>([&](auto e8) { constexpr int j = decltype(e8)::value; if (j < r.nv) cp[j] = v[j]; });
                    }
                } else {
                    float xres[8], xaux[8];
                    epilogue_operands(p, r, coff, t.pre[i][q], xres, xaux);
                    epilogue_row8(p, v, r, bz, coff, xres, xaux, cols[q]);
                }
            }
        }
    });
}

// After the MFMAs a lane owns output row c16 and 4 consecutive columns 4*g .. 4*g+3 of each 16x16 fragment:
// stored directly, a wavefront store would touch 16 rows x 32 bytes.  Instead the tile goes through LDS in
// bands of 32 rows (fragment row i of both wave rows), and every thread finishes 8 consecutive columns of
// one row: 16-byte bf16 accesses, whole 128-byte row segments per 8 lanes, for C, res, aux and pre_out alike.
// 256 threads = 2x2 waves of WM x WN; `band` = at least 32 x (BN + 4) floats of idle LDS.
template <int BN, int WM, int WN, int FM, int FN>
__device__ __forceinline__ void epilogue_tile(const toist_gemm& p, f32x4_t (&acc)[FM][FN], float* band, const int m0, const int n0,
                                              const int bz, const long long coff, const int ksl) {
    constexpr int CPR = EpiTile<BN, FM>::CPR, CH = EpiTile<BN, FM>::CH;
    // k-slice partial: raw f32 into the workspace, epilogue in splitk_reduce_kernel / splitk_epilogue_kernel
    const bool partial = p.split_k > 1;
    EpiCols cols[CH];
#pragma unroll
    for (int q = 0; q < CH; ++q)
        if (!partial) epilogue_cols(p, n0 + (((int)threadIdx.x + 256 * q) % CPR) * 8, cols[q], p.group ? p.group[bz].shift_off : 0);
    EpiTile<BN, FM> t;
    epilogue_request<BN, WM, FM>(p, m0, n0, coff, partial, t);
    epilogue_finish<BN, WM, WN, FM, FN>(p, acc, band, t, cols, bz, coff, ksl, partial);
}

// ---- lean epilogue -------------------------------------------------------------------------------------------------------
// epilogue_tile above serves every option of toist_epilogue, and the hot-path GEMMs were bound by ITS instruction stream: ~1000 VALU
// instructions per 64x64 tile per wave (64-bit row maps, per-option branches, spilled descriptor words read back lane by lane), 4
// cycles each on a wave64 -- 8400 cycles per tile measured inside the short-K kernel against ~500 of MFMA work.  The convolutions
// of the backbone need a fraction of it: bf16 rows in whole 16-byte chunks, per-column scale / shift, a residual, ReLU or a mask.
// lean_epilogue_ok() admits exactly that; everything else keeps the general path.
template <int BN, int WM, int WN, int FM, int FN>
__device__ __forceinline__ void epilogue_lean(const toist_gemm& p, f32x4_t (&acc)[FM][FN], float* band, const int m0, const int n0, const int bz,
                                              const long long coff, const int tid_ = -1) {
    constexpr int CPR = BN / 8, CH = (32 * CPR + 255) / 256, LDT = BN + 4;
    // tid_ >= 0: the calling workgroup is several 256-thread quads, each finishing its own 2x2-wave tile through its own band buffer
    const int tid = tid_ >= 0 ? tid_ : (int)threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, g = lane >> 4, c16 = lane & 15;
    const int M = p.M, N = p.N;
    const toist_epilogue& e = p.epi;
    bf16_t* const outp = (bf16_t*)p.c + coff;
    const int act = e.act;
    const bool masked = act >= TOIST_ACT_MASK_POS;         // activations that read aux
    const bf16_t* const resp = e.res ? (const bf16_t*)e.res + coff : nullptr;
    const bf16_t* const auxp = masked ? (const bf16_t*)e.aux + coff : nullptr;
    const int ldc = p.ldc, ldr = e.ldr, ldaux = e.ldaux;
    const float alpha = e.alpha;
    const int drop = e.drop_where;
    const unsigned long long dseed = drop ? e.drop_seed + (e.drop_seed_dev ? *e.drop_seed_dev : 0ull) : 0ull;
    const unsigned dth = (unsigned)(e.drop_p * 4294967296.0);
    const float dsc = 1.f / (1.f - e.drop_p);
    int rloc[CH], ncol[CH], brow[CH], c8[CH];
    bool col_ok[CH];
    float mul[CH][8], add[CH][8];
#pragma unroll
    for (int q = 0; q < CH; ++q) {
        const int c = tid + 256 * q;
        brow[q] = c / CPR; c8[q] = c - brow[q] * CPR;
        rloc[q] = (brow[q] >> 4) * WM + (brow[q] & 15);
        ncol[q] = n0 + c8[q] * 8;
        col_ok[q] = c < 32 * CPR && ncol[q] < N;            // N % 8 == 0: a chunk is whole or absent
#pragma unroll
        for (int j = 0; j < 8; ++j) { mul[q][j] = alpha; add[q][j] = 0.f; }
        if (col_ok[q]) {
            if (e.scale) {
                float t[8];
                load_cols8(e.scale + ncol[q], 8, t, 1.f);
#pragma unroll
                for (int j = 0; j < 8; ++j) mul[q][j] = alpha * t[j];
            }
            if (e.shift) load_cols8(e.shift + ncol[q], 8, add[q], 0.f);
        }
    }
    // output row of GEMM row m (the optional scatter map of a strided data gradient); residual / mask chunks of every band requested up front
    long long crow[FM][CH];
    uint4 rres[FM][CH], raux[FM][CH];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int q = 0; q < CH; ++q) {
            const int m = m0 + rloc[q] + 16 * i;
            crow[i][q] = m;
            if (col_ok[q] && m < M) {
                if (e.cmap) {
                    const int plane = e.cOH * e.cOW;
                    const int n_img = m / plane, rem = m - n_img * plane;
                    const int oy = rem / e.cOW, ox = rem - oy * e.cOW;
                    crow[i][q] = ((long long)n_img * e.cH + (long long)oy * e.cst) * e.cW + (long long)ox * e.cst;
                }
                if (resp) rres[i][q] = *reinterpret_cast<const uint4*>(resp + crow[i][q] * ldr + ncol[q]);
                if (auxp) raux[i][q] = *reinterpret_cast<const uint4*>(auxp + crow[i][q] * ldaux + ncol[q]);
            }
        }
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        lds_barrier();                // the last k-tile / the previous band is consumed
#pragma unroll
        for (int j = 0; j < FN; ++j) *reinterpret_cast<f32x4_t*>(band + (wm * 16 + c16) * LDT + wn * WN + j * 16 + g * 4) = acc[i][j];
        lds_barrier();
#pragma unroll
        for (int q = 0; q < CH; ++q) {
            const int m = m0 + rloc[q] + 16 * i;
            if (col_ok[q] && m < M) {
                const f32x4_t lo = *reinterpret_cast<const f32x4_t*>(band + brow[q] * LDT + c8[q] * 8);
                const f32x4_t hi = *reinterpret_cast<const f32x4_t*>(band + brow[q] * LDT + c8[q] * 8 + 4);
                float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = v[j] * mul[q][j] + add[q][j];
                const unsigned long long didx = ((unsigned long long)bz * M + m) * N + ncol[q];     // same element index as epilogue_row8
                if (drop == 1) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = dropout_keep(dseed, didx + j, dth) ? v[j] * dsc : 0.f;
                }
                if (resp) {
                    float x[8];
                    unpack8(rres[i][q], x);
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] += x[j];
                }
                if (act == TOIST_ACT_RELU) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
                } else if (act == TOIST_ACT_GELU) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = gelu_f(v[j]);
                } else if (masked) {
                    float x[8];
                    unpack8(raux[i][q], x);
                    if (act == TOIST_ACT_MASK_POS) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) v[j] = x[j] > 0.f ? v[j] : 0.f;
                    } else if (act == TOIST_ACT_GELU_BWD) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) v[j] *= gelu_grad_f(x[j]);
                    } else {
#pragma unroll
                        for (int j = 0; j < 8; ++j) v[j] *= x[j] * (1.f - x[j]);
                    }
                }
                if (drop == 2) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = dropout_keep(dseed, didx + j, dth) ? v[j] * dsc : 0.f;
                }
                *reinterpret_cast<uint4*>(outp + crow[i][q] * ldc + ncol[q]) =
                    make_uint4(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7]));
            }
        }
    }
}

// One output tile (tile `tl` of the launch's padded, XCD-striped tile list) of one (batch, k-slice) problem.
template <int BM, int BN, int BK, int AK, int BKD, int NS, bool LEAN>
__device__ __forceinline__ void gemm_tile(const toist_gemm& p, const int tl, bf16_t* const smem, const int z, const bool direct) {
    constexpr int WM = BM / 2, WN = BN / 2, FM = WM / 16, FN = WN / 16;
    constexpr int ACH = BM * BK / 8 / 256, BCH = BN * BK / 8 / 256;  // 1 KiB DMA pieces per wave per tile
    constexpr bool A_KM = (AK == TOIST_A_KROW);    // A staged k-major
    constexpr bool B_KM = (BKD != TOIST_B_ROWK);   // B staged k-major
    constexpr int SA_ELEMS = BM * BK, SB_ELEMS = BN * BK;
    constexpr int STAGE = SA_ELEMS + SB_ELEMS;                       // elements per ring slot
    // NS = slots of the DMA ring (NS-1 k-tiles in flight per workgroup).  What the kernel can pull from L2 is
    // (bytes in flight per CU) / (loaded latency, ~1.4 us): the host picks NS and the tile so that the whole grid
    // is resident at once with as many staged bytes as the 160 KB of LDS allow (see pick_tile below).
    constexpr int CNT = ACH + BCH;
    static_assert(NS >= 2 && NS <= 4, "wait ladder below covers up to 2 younger tiles");

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int g = lane >> 4, c16 = lane & 15;

    const int M = p.M, N = p.N, K = p.K;
    // XCD-aware tile order: the dispatcher deals workgroups round-robin over the 8 XCDs (id % 8), each with its own
    // 4 MB L2.  Tile L = (id % 8) * ceil(tiles / 8) + id / 8 gives every XCD one contiguous run of tiles, visited in
    // panels of 8 M rows (tile_order): the workgroups resident on an XCD at one time cover ~8 M tiles x a few N tiles,
    // so both operands are re-used from ONE L2 -- the N tiles of an M row share their A tile, neighbouring M tiles share
    // the halo rows of a 3x3 gather -- instead of every XCD streaming a whole operand through its L2.
    const int nt_n = (p.N + BN - 1) / BN;
    const int nt_m = (p.M + BM - 1) / BM;
    const int tiles = nt_m * nt_n;
    const int tile_id = direct ? tl : (tl & 7) * ((tiles + 7) >> 3) + (tl >> 3);
    if (tile_id >= tiles) return;            // the tile list is padded to a multiple of 8
    int tile_m, tile_n;
    tile_order(tile_id, nt_m, nt_n, tile_m, tile_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int bz = z / p.split_k, ksl = z - bz * p.split_k;
    const int bo = bz / p.batch_inner, bi = bz - bo * p.batch_inner;
    const toist_operand oa = p.a, ob = p.b;
    const bf16_t* a_base = (const bf16_t*)oa.ptr + bo * oa.bs_outer + bi * oa.bs_inner;
    const bf16_t* b_base = (const bf16_t*)ob.ptr + bo * ob.bs_outer + bi * ob.bs_inner;
    long long coff_ = bo * p.cs_outer + bi * p.cs_inner;
    if (p.group) {   // grouped launch: per-problem base pointers from the table (uniform loads)
        const toist_group gq = p.group[bz];
        a_base = (const bf16_t*)gq.a;
        b_base = (const bf16_t*)gq.b;
        coff_ = gq.c_off;
    }
    if (AK == TOIST_A_ROWK && p.a2 != nullptr && n0 >= p.a2_from)      // second A operand for the upper output columns (packed in_proj)
        a_base = (const bf16_t*)p.a2 + bo * oa.bs_outer + bi * oa.bs_inner;
    const i32x4_t rsA = make_rsrc(a_base);
    const i32x4_t rsB = make_rsrc(b_base);
    const long long coff = coff_;
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
    const bool want_csum = A_KM && p.a_colsum != nullptr && tile_n == 0;
#pragma unroll
    for (int it = 0; it < ACH; ++it)
#pragma unroll
        for (int j = 0; j < 8; ++j) csum[it][j] = 0.f;

    // ---- walker state of the issue stream (next tile to issue = kt_issue) --------------------------------------
    constexpr bool A_GATHER = (AK == TOIST_A_CONV || AK == TOIST_A_CONVT);
    const bool a_fast = A_GATHER && (oa.SC % BK) == 0;
    const bool b_two = (BKD == TOIST_B_KROW) && ob.kin > 0;
    const bool b_fast = b_two && (ob.kin % BK) == 0;
    int k_issue = kt_beg * BK;
    TapPos ta, tb;
    ta.tap = ta.inner = ta.r = ta.s = 0; tb = ta;
    int a_off[ACH]; bool a_in[ACH];          // gathers: per-tap offset/validity; KROW: running offset
    int b_off[BCH]; bool b_in[BCH];
    PixPos bp[BCH];
    int b_u = 0;                             // KROW two-level k: uniform part of the offset
#pragma unroll
    for (int it = 0; it < ACH; ++it) { a_off[it] = 0; a_in[it] = false; }
#pragma unroll
    for (int it = 0; it < BCH; ++it) { b_off[it] = 0; b_in[it] = false; bp[it].n = bp[it].py = bp[it].px = 0; }
    if (A_GATHER && a_fast) {
        ta.init(k_issue, oa.SC, oa.S);
#pragma unroll
        for (int it = 0; it < ACH; ++it) conv_tap<AK>(ca[it], oa, ta, a_off[it], a_in[it]);
    }
    if (AK == TOIST_A_KROW) {
#pragma unroll
        for (int it = 0; it < ACH; ++it) a_off[it] = ca[it].base + (k_issue + ca[it].row) * lda;
    }
    if (BKD == TOIST_B_KROW) {
        if (b_fast) { tb.init(k_issue, ob.kin, 1); b_u = tb.inner * ldb + tb.tap * (int)ob.tap_stride; }
#pragma unroll
        for (int it = 0; it < BCH; ++it) b_off[it] = cb[it].base + cb[it].row * ldb + (b_two ? 0 : k_issue * ldb);
    }
    const int cx_dq = (BKD == TOIST_B_CONVX) ? BK / ob.PW : 0, cx_dr = (BKD == TOIST_B_CONVX) ? BK - cx_dq * ob.PW : 0;
    if (BKD == TOIST_B_CONVX) {
#pragma unroll
        for (int it = 0; it < BCH; ++it) bp[it].init(k_issue + cb[it].row, ob.PH, ob.PW);
    }

    const unsigned lds0 = (unsigned)(size_t)smem;  // LDS byte address of the ring
    auto issue = [&](int slot) {                   // stages tile k_issue / BK into `slot`, then advances the walkers
        const unsigned sbase = lds0 + (unsigned)slot * (STAGE * 2) + (unsigned)wave * 1024u;
        const int k0 = k_issue;
#pragma unroll
        for (int it = 0; it < ACH; ++it) {
            const unsigned dst = sbase + it * 4096u;
            if (AK == TOIST_A_ROWK) dma16(dst, rsA, ca[it].base + k0, ca[it].ok && (k0 + ca[it].kc * 8 < K));
            else if (AK == TOIST_A_KROW) { dma16(dst, rsA, a_off[it], ca[it].ok && (k0 + ca[it].row < K)); a_off[it] += BK * lda; }
            else if (a_fast) dma16(dst, rsA, a_off[it] + ta.inner, a_in[it]);
            else load_a_slow<AK, BK>(dst, rsA, ca[it], oa, k0);
        }
#pragma unroll
        for (int it = 0; it < BCH; ++it) {
            const unsigned dst = sbase + SA_ELEMS * 2 + it * 4096u;
            if (BKD == TOIST_B_ROWK) dma16(dst, rsB, cb[it].base + k0, cb[it].ok && (k0 + cb[it].kc * 8 < K));
            else if (BKD == TOIST_B_KROW) {
                const int k = k0 + cb[it].row;
                if (!b_two) { dma16(dst, rsB, b_off[it], cb[it].ok && k < K); b_off[it] += BK * ldb; }
                else if (b_fast) dma16(dst, rsB, b_off[it] + b_u, cb[it].ok && k < K);
                else {   // a k-tile straddles taps: per-row tap
                    const int tap = k / ob.kin;
                    dma16(dst, rsB, cb[it].base + (k - tap * ob.kin) * ldb + tap * (int)ob.tap_stride, cb[it].ok && k < K);
                }
            } else {     // CONVX: k = output pixel, columns = (tap, c)
                const PixPos& q = bp[it];
                const int iy = q.py * ob.stride - ob.pad + cb[it].r * ob.dil, ix = q.px * ob.stride - ob.pad + cb[it].s * ob.dil;
                const bool in = cb[it].ok && (k0 + cb[it].row < K) && iy >= 0 && ix >= 0 && iy < ob.SH && ix < ob.SW;
                dma16(dst, rsB, cb[it].base + ((q.n * ob.SH + iy) * ob.SW + ix) * ob.SC, in);
                bp[it].advance(cx_dq, cx_dr, ob.PH, ob.PW);
            }
        }
        k_issue += BK;
        if (A_GATHER && a_fast) {
            if (ta.advance(BK, oa.SC, oa.S)) {
#pragma unroll
                for (int it = 0; it < ACH; ++it) conv_tap<AK>(ca[it], oa, ta, a_off[it], a_in[it]);
            }
        }
        if (BKD == TOIST_B_KROW && b_fast) {
            tb.advance(BK, ob.kin, 1);
            b_u = tb.inner * ldb + tb.tap * (int)ob.tap_stride;
        }
    };

    const int ntiles = (p.flags & 256) ? 0 : kt_end - kt_beg;     // flags bit 8 (experiments): skip the reduction loop
#pragma unroll
    for (int t = 0; t < NS - 1; ++t)
        if (t < ntiles) issue(t);

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
            issue(ns);  // next tile in order; refills the slot tile t-1 was read from
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

    if (want_csum) {
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
            if (m0 + col < M) atomicAdd(p.a_colsum + (p.group ? p.group[bz].colsum_off : 0) + m0 + col, t);
        }
    }

    static_assert(32 * (BN + 4) * 4 <= NS * STAGE * 2, "epilogue band must fit the (now idle) ring");
    if (p.flags & 512) return;                                    // flags bit 9 (experiments): skip the epilogue
    if constexpr (LEAN) epilogue_lean<BN, WM, WN, FM, FN>(p, acc, reinterpret_cast<float*>(smem), m0, n0, bz, coff);
    else epilogue_tile<BN, WM, WN, FM, FN>(p, acc, reinterpret_cast<float*>(smem), m0, n0, bz, coff, ksl);
}

// Persistent launch: the hardware dispatches ~530 workgroups per microsecond chip-wide (measured: a 12800-tile launch whose
// workgroups return at once takes 24 us, a 3200-tile one 6.9 us), which for the K <= 256 GEMMs of the hot path is as much as their
// whole reduction loop.  The grid is therefore capped at what the chip holds at once (launch_variant) and every workgroup walks
// the tile list with stride gridDim.x; a multiple of 8, so a workgroup stays on its XCD's contiguous run of tiles.
// Waves per SIMD asked of the register allocator.  Round 2's descriptor options (second A operand, groups, the persistent loop, the
// up-front residual requests) had grown the 64x64 kernels from 97 + 16 to 128 + 16 registers and the 128x64 / 64x128 ones from 125 + 32
// to 158-186 + 32: one occupancy step lost each (round-1 build, same box: 4096^3 on 128x64 tiles 726 TFLOP/s, now 672).
#ifndef GEMM_WAVES_64
#define GEMM_WAVES_64 4
#endif
#ifndef GEMM_WAVES_128
#define GEMM_WAVES_128 3
#endif
template <int BM, int BN, int NS>
constexpr int gemm_min_waves() { return (BM * BN <= 4096 && NS == 2) ? GEMM_WAVES_64 : (BM * BN == 8192 ? GEMM_WAVES_128 : 1); }

template <int BM, int BN, int BK, int AK, int BKD, int NS, bool LEAN = false>
__global__ __launch_bounds__(256, (gemm_min_waves<BM, BN, NS>())) void gemm_kernel(const toist_gemm p) {
    constexpr int STAGE = (BM + BN) * BK;
    __shared__ __attribute__((aligned(16))) bf16_t smem[NS * STAGE];
    const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    if (p.flags & 1024) {
        // Grouped launch (flag set by launch_variant): whole problems are pinned to XCDs.  Workgroup L runs on XCD L % 8 (dispatch order); XCD x walks the problems
        // x, x + 8, ... one after the other, all tiles of a problem side by side on that XCD's CUs, so a problem's operands are pulled
        // into ONE L2.  (Striping every problem's tiles over all eight XCDs, as for a single problem, made each XCD stream every
        // problem: the 22 grouped layer-3 3x3 weight gradients fetched 9 x their operands -- profiles/r02_pmc_fetch_summary.txt.)
        const int L = (int)blockIdx.x, q = (L >> 3) / tiles, t = (L >> 3) - q * tiles, z = (L & 7) + 8 * q;
        if (z < p.batch * p.split_k) gemm_tile<BM, BN, BK, AK, BKD, NS, LEAN>(p, t, smem, z, true);
        return;
    }
    const int tiles8 = (tiles + 7) & ~7;
    for (int tl = (int)blockIdx.x; tl < tiles8; tl += (int)gridDim.x) {
        gemm_tile<BM, BN, BK, AK, BKD, NS, LEAN>(p, tl, smem, (int)blockIdx.z, false);
        if (tl + (int)gridDim.x < tiles8) __syncthreads();       // the next tile's DMA reuses the LDS the epilogue bands lived in
    }
}

// ---- 3x3 / stride 1 / pad 1 convolution with a shared halo patch -------------------------------------------------
// The implicit GEMM above stages every tap's A tile separately: the same input pixels travel L2 -> LDS nine times,
// and that transfer (LDS-DMA issue rate, ~35 B/clk/CU) is what bounds the kernel.  Here a workgroup computes
// BM = 128 consecutive (flattened NHWC) pixels x 64 output channels and, per 64-channel chunk of the reduction,
// stages ONE patch of BM + 2(W+1) pixel rows; the nine taps read their A fragments from that patch at row offset
// dy*W + dx (rows that fall outside the image are zeroed in registers).  Only the 8 KB weight tile changes per tap:
// 100 KB of DMA per (chunk, 9 taps) instead of 216 KB for the same tile through the generic kernel, 288 KB at 64x64.
//   forward  (A_CONV,  B_ROWK): src = x  [N,H,W,C],  dy = r-1, dx = s-1, B rows = output channels, k-contiguous
//   dgrad    (A_CONVT, B_KROW): src = dy [N,H,W,Co], dy = 1-r, dx = 1-s, B tile k-major (k = co, columns = c)
// Pipeline: patch double-buffered per chunk (its 28 pieces are issued one per wave during the first 7 taps of the
// previous chunk), weight tiles in a 3-slot ring, one raw s_barrier per (chunk, tap) step, hand-counted vmcnt.
constexpr int C3_BM = 128, C3_BN = 64, C3_BK = 64, C3_NP = 28, C3_NB = 3;
constexpr int C3_PATCH = C3_NP * 512;                 // elements per patch buffer (28 KiB)
constexpr int C3_BT = C3_BN * C3_BK;                  // elements per weight tile (8 KiB)
constexpr int C3_LDS = (2 * C3_PATCH + C3_NB * C3_BT) * 2;   // 80 KiB: two workgroups per CU

template <bool DGRAD, bool LEAN>
__global__ __launch_bounds__(256) void conv3_kernel(const toist_gemm p) {
    constexpr int BM = C3_BM, BN = C3_BN, BK = C3_BK, WM = 64, WN = 32, FM = 4, FN = 2;
    extern __shared__ __attribute__((aligned(16))) bf16_t smem[];
    bf16_t* const patch = smem;
    bf16_t* const btile = smem + 2 * C3_PATCH;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, g = lane >> 4, c16 = lane & 15;
    const int M = p.M, N = p.N;
    const toist_operand oa = p.a, ob = p.b;
    const int W = oa.SW, H = oa.SH, C = oa.SC;        // source plane and channels (= reduction width per tap)
    const int nt_n = (N + BN - 1) / BN, nt_m = (M + BM - 1) / BM;
    const int tiles = nt_m * nt_n;
    const int tile_id = (int)(blockIdx.x & 7) * ((tiles + 7) >> 3) + (int)(blockIdx.x >> 3);   // XCD-aware order (see gemm_kernel)
    if (tile_id >= tiles) return;
    int tile_m, tile_n;
    tile_order(tile_id, nt_m, nt_n, tile_m, tile_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const i32x4_t rsA = make_rsrc(oa.ptr), rsB = make_rsrc(ob.ptr);
    const int halo = W + 1;
    const int nchunks = C / BK, nsteps = nchunks * 9;

    // ---- per-lane tap validity of the FM output rows this lane feeds into the MFMAs ----
    unsigned vmask[FM];
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const int m = m0 + wm * WM + i * 16 + c16;
        const int rem = m % (H * W);
        const int y = rem / W, x = rem - y * W;
        unsigned v = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int r = t / 3, s_ = t - r * 3;
            const int dy = DGRAD ? 1 - r : r - 1, dx = DGRAD ? 1 - s_ : s_ - 1;
            if (m < M && y + dy >= 0 && y + dy < H && x + dx >= 0 && x + dx < W) v |= 1u << t;
        }
        vmask[i] = v;
    }

    // wave-uniform: taps for which fragment row i needs no masking at all (every lane valid) -> the select is skipped
    unsigned clean[FM];
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        unsigned c = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t)
            if (__builtin_amdgcn_ballot_w64((vmask[i] >> t) & 1u) == ~0ull) c |= 1u << t;
        clean[i] = __builtin_amdgcn_readfirstlane(c);
    }

    // ---- lane invariants of the DMA pieces and of the fragment reads (everything that varies per step is wave-uniform) ----
    const unsigned lds0 = (unsigned)(size_t)smem;
    // patch piece q = tap*4 + wave of a chunk covers patch rows q*8 + (lane >> 3); its swizzle does not depend on the tap
    const int p_row0 = wave * 8 + (lane >> 3);                              // + tap*32
    const int p_kc = (lane & 7) ^ ((((wave & 1) << 2) + (lane >> 4)) & 7);
    const int p_pix0 = m0 - halo + p_row0;                                  // + tap*32
    const int p_off0 = p_pix0 * C + p_kc * 8;                               // + tap*32*C + chunk*BK
    const int p_rows = BM + 2 * halo;
    // weight-tile pieces (2 per wave)
    int b_off0[2];
    bool b_ok[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int pch = (it * 4 + wave) * 64 + lane;
        if (DGRAD) {   // k-major tile [BK k = co][BN n = c]
            const int krow = pch / (BN / 8), rc = swz_m<BN>(krow, pch % (BN / 8));
            const int nn = n0 + rc * 8;
            b_off0[it] = nn + krow * ob.ld;                                 // + chunk*BK*ld + tap*tap_stride
            b_ok[it] = nn < N;
        } else {       // k-contiguous tile [BN n = co][BK k = c]
            const int row = pch >> 3, kc = swz_k<BK>(row, pch & 7);
            b_off0[it] = (n0 + row) * ob.ld + kc * 8;                        // + tap*C + chunk*BK
            b_ok[it] = (n0 + row) < N;
        }
    }
    const int b_chunk_step = DGRAD ? BK * ob.ld : BK, b_tap_step = DGRAD ? (int)ob.tap_stride : C;
    // fragment reads: rows 16 apart share their swizzle, so fragments i / j are constant byte offsets from one address
    const int a_row0 = wm * WM + c16;                                       // + roff(tap)
    int b_frag[2][FN][2];                                                   // [ks][j][lo/hi] element offsets inside a weight tile
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            if (DGRAD) {   // k-major tile: the swizzle mixes the row chunk, so every j has its own pair of addresses
                const int k = ks * 32 + 8 * g + (c16 >> 2);
                const int rc = ((wn * WN + j * 16) >> 3) + ((c16 & 3) >> 1), sub = (c16 & 1) * 4;
                b_frag[ks][j][0] = k * BN + swz_m<BN>(k, rc) * 8 + sub;
                b_frag[ks][j][1] = (k + 4) * BN + swz_m<BN>(k + 4, rc) * 8 + sub;
            } else {
                const int row = wn * WN + j * 16 + c16;
                b_frag[ks][j][0] = row * BK + swz_k<BK>(row, ks * 4 + g) * 8;
                b_frag[ks][j][1] = 0;
            }
        }

    auto issue_patch = [&](int chunk, int tap) {                            // patch piece (tap, wave) of `chunk`
        const int row = p_row0 + tap * 32, pix = p_pix0 + tap * 32;
        const bool ok = pix >= 0 && pix < M && row < p_rows;
        dma16(lds0 + (unsigned)(chunk & 1) * (C3_PATCH * 2) + (unsigned)(tap * 4 + wave) * 1024u, rsA, p_off0 + tap * 32 * C + chunk * BK, ok);
    };
    auto issue_b = [&](int slot, int chunk, int tap) {
        const unsigned dst = lds0 + (unsigned)(2 * C3_PATCH + slot * C3_BT) * 2u + (unsigned)wave * 1024u;
        const int u = chunk * b_chunk_step + tap * b_tap_step;
        dma16(dst, rsB, b_off0[0] + u, b_ok[0]);
        dma16(dst + 4096u, rsB, b_off0[1] + u, b_ok[1]);
    };

    f32x4_t acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // prologue: whole patch of chunk 0 (7 pieces per wave), weight tiles of steps 0 and 1
#pragma unroll
    for (int q = 0; q < 7; ++q) issue_patch(0, q);
    issue_b(0, 0, 0);
    issue_b(1, 0, 1);                                                       // nsteps >= 9

    int chunk = 0, tap = 0, slot = 0;                                       // step t = chunk*9 + tap reads weight slot t % 3
    int ichunk = 0, itap = 2, islot = 2;                                    // weight tile t + 2 (the next one to issue)
    bool prev_patch = false;                                                // did step t-1 issue a patch piece?
    for (int t = 0; t < nsteps; ++t) {
        // loads this wave issued after weight tile t: step t-1's group = [patch piece?] + tile t+1
        if (t + 1 >= nsteps) wait_vm<0>();
        else if (prev_patch) wait_vm<3>();
        else wait_vm<2>();
        __builtin_amdgcn_s_barrier();
        // group(t): next chunk's patch piece (first 7 taps), then weight tile t+2 -- refills the slot step t-1 read
        prev_patch = tap < 7 && chunk + 1 < nchunks;
        if (prev_patch) issue_patch(chunk + 1, tap);
        if (t + 2 < nsteps) issue_b(islot, ichunk, itap);
        if (++itap == 9) { itap = 0; ++ichunk; }
        if (++islot == C3_NB) islot = 0;

        const bf16_t* sP = patch + (chunk & 1) * C3_PATCH;
        const bf16_t* sB = btile + slot * C3_BT;
        const int r = (tap >= 6) ? 2 : (tap >= 3 ? 1 : 0), s_ = tap - r * 3;
        const int arow = a_row0 + halo + (DGRAD ? (1 - r) * W + (1 - s_) : (r - 1) * W + (s_ - 1));
        const int asw = (arow >> 1) & 7;
#pragma unroll
        for (int ks = 0; ks < BK / 32; ++ks) {
            bf16x8_t af[FM], bfr[FN];
            const bf16_t* ap = sP + arow * BK + ((ks * 4 + g) ^ asw) * 8;
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                af[i] = *reinterpret_cast<const bf16x8_t*>(ap + i * 16 * BK);
                if (!((clean[i] >> tap) & 1u)) {                            // wave-uniform: some lane of this fragment is outside the image
                    const bf16x8_t zero = {0, 0, 0, 0, 0, 0, 0, 0};
                    af[i] = ((vmask[i] >> tap) & 1u) ? af[i] : zero;
                }
            }
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                if (DGRAD) {
                    typedef __attribute__((address_space(3))) s16x4_t* lds_v4;
                    union { struct { s16x4_t a, b; } h; bf16x8_t v; } u;
                    u.h.a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(sB + b_frag[ks][j][0]));
                    u.h.b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(sB + b_frag[ks][j][1]));
                    bfr[j] = u.v;
                } else {
                    bfr[j] = *reinterpret_cast<const bf16x8_t*>(sB + b_frag[ks][j][0]);
                }
            }
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
        }
        if (++tap == 9) { tap = 0; ++chunk; }
        if (++slot == C3_NB) slot = 0;
    }
    wait_vm<0>();
    if constexpr (LEAN) epilogue_lean<BN, WM, WN, FM, FN>(p, acc, reinterpret_cast<float*>(smem), m0, n0, 0, 0);
    else epilogue_tile<BN, WM, WN, FM, FN>(p, acc, reinterpret_cast<float*>(smem), m0, n0, 0, 0, 0);
}


// ---- 128 x 128 tiles with 64 x 64 wave tiles for deep row-major GEMMs (round 3) ---------------------------------------------------------
// The 1x1 convolutions with K >= 512 (ResNet conv1 of every bottleneck and the data gradient of conv3: 12800 x 256 x 1024 in layer 3,
// 45 launches per step at 20 - 24 us) run on 64 x 64 tiles, which are bound by the texture path: every k-tile of a workgroup stages
// 16 KB for 8 MFMAs per wave (800 tiles x 16 k-tiles x 16 KB = 205 MB through 256 CUs at <= 64 B/clk each), and a 32 x 32 wave tile
// reads 1 KiB of LDS operands per MFMA.  What the experiments of this round say a kernel needs (profiles/r03_conv3_variants.txt,
// r03_panel2_phase_cycles.txt): (1) 64 x 64 outputs per wave (0.5 KiB of LDS reads per MFMA); (2) two waves per SIMD from ONE barrier
// domain -- here the two k-halves of every 64-deep k-tile go to two waves of a SIMD pair, their partial sums are exchanged once at the
// end; (3) few instructions per MFMA: the lane offsets of the DMA pieces are loop invariants, the k offset travels in the buffer
// instruction's scalar offset, ring slots are compile-time (4-step unrolled loop), fragments are read one k-tile ahead; (4) no register
// spills (128 accumulator + fragment registers, ~40 for everything else) and no LDS band in the epilogue (fragments are finished where
// the MFMA left them, 8-byte accesses).  Staged bytes per MFMA: half of the 64 x 64 tiling.
#ifndef GEMM_UNIT   // main translation unit only
constexpr int G8_BM = 128, G8_BN = 128, G8_BK = 64;
constexpr int G8_TILE = G8_BM * G8_BK;                 // elements of one operand tile (16 KiB)
constexpr int G8_STAGE_BYTES = 2 * G8_TILE * 2;        // A + B

struct G8Frags { bf16x8_t a[4]; bf16x8_t b[4]; };

template <int AK, int BKD, int NS>   // NS = slots of the DMA ring (4 or 5: 128 / 160 KiB of LDS, one workgroup per CU either way)
__global__ __launch_bounds__(512, 2) void gemm128_kernel(const toist_gemm p) {
    constexpr int BM = G8_BM, BN = G8_BN, BK = G8_BK, WM = 64, WN = 64, FM = 4, FN = 4;
    constexpr bool B_KM = BKD == TOIST_B_KROW;
    constexpr bool GATHER = AK != TOIST_A_ROWK;          // stride-1 convolution gather (A_CONV) / its transposed gather (A_CONVT, same plane size)
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kh = wave >> 2, wm = (wave >> 1) & 1, wn = wave & 1, g = lane >> 4, c16 = lane & 15;
    const int M = p.M, N = p.N, K = p.K;
    const int nt_n = (N + BN - 1) / BN, nt_m = (M + BM - 1) / BM;
    const int tiles = nt_m * nt_n;
    const int tile_id = (int)(blockIdx.x & 7) * ((tiles + 7) >> 3) + (int)(blockIdx.x >> 3);
    if (tile_id >= tiles) return;
    int tile_m, tile_n;
    tile_order(tile_id, nt_m, nt_n, tile_m, tile_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const toist_operand oa = p.a;
    const int lda = oa.ld, ldb = p.b.ld;
    const int T = K / BK;                               // k-tiles (K % 64 == 0)
    const int ntaps = GATHER ? oa.R * oa.S : 1, nch = T / ntaps;        // k = tap * (nch * 64) + channel
    // gathers: the descriptor base is moved to the source pixel of tap (0, 0) of output pixel (0, 0, 0), so that lane offsets (pixel)
    // and tap offsets (scalar) are both non-negative.  A_CONVT walks the taps in reverse: source = p + pad - (R-1) dil + r' dil.
    const int sh_y = GATHER ? (AK == TOIST_A_CONV ? -oa.pad : oa.pad - (oa.R - 1) * oa.dil) : 0;
    const int sh_x = GATHER ? (AK == TOIST_A_CONV ? -oa.pad : oa.pad - (oa.S - 1) * oa.dil) : 0;
    const i32x4_t rsA = make_rsrc(GATHER ? (const void*)((const bf16_t*)oa.ptr + ((long long)sh_y * oa.SW + sh_x) * oa.SC) : oa.ptr);
    const i32x4_t rsB = make_rsrc(p.b.ptr);
    const unsigned lds0 = (unsigned)(size_t)lds_raw;
    const bool timing = (p.flags & 2048) && p.workspace != nullptr;     // experiments: shader-clock stamps of prologue / loop / epilogue
    unsigned long long T0 = 0, T1 = 0, T2 = 0;
    if (timing) T0 = cyc_now();

    // ---- DMA pieces: two of A and two of B per wave and k-tile; lane offsets in bytes (or out of range), k in the scalar offset ----
    int va[2], vb[2];
    unsigned tapok[2] = {0xffffu, 0xffffu};             // gathers: bit t = this lane's source pixel of tap t lies inside the plane
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int pch = (it * 8 + wave) * 64 + lane;
        const int row = pch >> 3, kc = swz_k<BK>(row, pch & 7);
        const int m = m0 + row;
        if (GATHER) {
            const int plane = oa.PH * oa.PW;
            const int mm = m < M ? m : 0;
            const int n = mm / plane, rem = mm - n * plane;
            const int py = rem / oa.PW, px = rem - py * oa.PW;
            const int st = AK == TOIST_A_CONV ? oa.stride : 1;         // strided forward gathers; the transposed gather is stride 1 only
            va[it] = m < M ? (((n * oa.SH + py * st) * oa.SW + px * st) * oa.SC + kc * 8) * 2 : OOB;
            unsigned ok = 0;
            for (int t = 0; t < ntaps; ++t) {
                const int r = t / oa.S, s_ = t - r * oa.S;
                const int iy = py * st + sh_y + r * oa.dil, ix = px * st + sh_x + s_ * oa.dil;
                if (iy >= 0 && iy < oa.SH && ix >= 0 && ix < oa.SW) ok |= 1u << t;
            }
            tapok[it] = ok;
        } else {
            va[it] = (m < M) ? (m * lda + kc * 8) * 2 : OOB;
        }
        if (B_KM) {    // k-major tile [BK k][BN n]
            const int krow = pch / (BN / 8), rc = swz_m<BN>(krow, pch % (BN / 8));
            const int nn = n0 + rc * 8;
            vb[it] = nn < N ? (nn + krow * ldb) * 2 : OOB;
        } else {
            vb[it] = (n0 + row < N) ? ((n0 + row) * ldb + kc * 8) * 2 : OOB;
        }
    }
    // issue-side walker over the k-tiles: (tap, 64-channel chunk)
    int i_tap = 0, i_chunk = 0;
    int va_eff[2] = {(tapok[0] & 1u) ? va[0] : OOB, (tapok[1] & 1u) ? va[1] : OOB};
    int so_tap = 0;                                     // bytes: source offset of the current tap (gathers)
    const long long b_tap = p.b.tap_stride;
    auto issue = [&](const int slot) {                  // all four pieces of the next k-tile in one statement (M0 saved once)
        const unsigned da = lds0 + (unsigned)(slot * G8_STAGE_BYTES) + (unsigned)wave * 1024u, db = da + (unsigned)(G8_TILE * 2);
        const int soa = so_tap + i_chunk * (BK * 2);
        int sob;
        if (B_KM) sob = (i_chunk * BK * ldb + (GATHER ? (ntaps - 1 - i_tap) * (int)b_tap : 0)) * 2;   // A_CONVT walks the taps in reverse: weight tap = last - t'
        else sob = (i_tap * nch + i_chunk) * (BK * 2);
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %3, %7, %9 offen lds\n\t"
                     "s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %4, %7, %9 offen lds\n\t"
                     "s_mov_b32 m0, %11\n\ts_nop 0\n\tbuffer_load_dwordx4 %5, %8, %10 offen lds\n\t"
                     "s_mov_b32 m0, %12\n\ts_nop 0\n\tbuffer_load_dwordx4 %6, %8, %10 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "s"(da), "s"(da + 8192u), "v"(va_eff[0]), "v"(va_eff[1]), "v"(vb[0]), "v"(vb[1]), "s"(rsA), "s"(rsB), "s"(__builtin_amdgcn_readfirstlane(soa)),
                       "s"(__builtin_amdgcn_readfirstlane(sob)), "s"(db), "s"(db + 8192u)
                     : "memory");
        if (++i_chunk == nch) {                         // next tap: its scalar offset and this lane's validity
            i_chunk = 0;
            ++i_tap;
            if (GATHER && i_tap < ntaps) {
                const int r = i_tap / oa.S, s_ = i_tap - r * oa.S;
                so_tap = ((r * oa.dil) * oa.SW + s_ * oa.dil) * oa.SC * 2;
#pragma unroll
                for (int it = 0; it < 2; ++it) va_eff[it] = ((tapok[it] >> i_tap) & 1u) ? va[it] : OOB;
            }
        }
    };
#pragma unroll
    for (int i = 0; i < NS - 1; ++i) issue(i);

    // ---- fragment addresses (bytes inside a stage): rows 16 apart share their swizzle -> one base + immediates ----
    const int a_row = wm * WM + c16;
    const int a_base = (a_row * BK + swz_k<BK>(a_row, kh * 4 + g) * 8) * 2;
    int b_base[FN][2];
#pragma unroll
    for (int j = 0; j < FN; ++j) {
        if (B_KM) {
            const int k = kh * 32 + 8 * g + (c16 >> 2);
            const int rc = ((wn * WN + j * 16) >> 3) + ((c16 & 3) >> 1), sub = (c16 & 1) * 4;
            b_base[j][0] = G8_TILE * 2 + (k * BN + swz_m<BN>(k, rc) * 8 + sub) * 2;
            b_base[j][1] = G8_TILE * 2 + ((k + 4) * BN + swz_m<BN>(k + 4, rc) * 8 + sub) * 2;
        } else {
            const int row = wn * WN + j * 16 + c16;
            b_base[j][0] = G8_TILE * 2 + (row * BK + swz_k<BK>(row, kh * 4 + g) * 8) * 2;
            b_base[j][1] = 0;
        }
    }
    auto load_frags = [&](G8Frags& f, auto slotc) {
        constexpr int SLOT = decltype(slotc)::value;
#pragma unroll
        for (int i = 0; i < FM; ++i) f.a[i] = *reinterpret_cast<const bf16x8_t*>(lds_raw + a_base + SLOT * G8_STAGE_BYTES + i * 16 * BK * 2);
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            if (B_KM) {
                typedef __attribute__((address_space(3))) s16x4_t* lds_v4;
                union { struct { s16x4_t a, b; } h; bf16x8_t v; } u;
                u.h.a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(lds_raw + b_base[j][0] + SLOT * G8_STAGE_BYTES));
                u.h.b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(lds_raw + b_base[j][1] + SLOT * G8_STAGE_BYTES));
                f.b[j] = u.v;
            } else {
                f.b[j] = *reinterpret_cast<const bf16x8_t*>(lds_raw + b_base[j][0] + SLOT * G8_STAGE_BYTES);
            }
        }
    };

    f32x4_t acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    wait_vm<4 * (NS - 2)>();                            // k-tile 0 landed (the younger ones may fly; K holds at least NS k-tiles)
    __builtin_amdgcn_s_barrier();
    G8Frags fa, fb;
    load_frags(fa, std::integral_constant<int, 0>{});

    if (timing) T1 = cyc_now();
    int t = 0;
    // one k-tile: `cur` = its fragments (read during the previous step), `nxt` receives those of k-tile t + 1 from ring slot SLOT + 1
    auto step = [&](auto slotc, G8Frags& cur, G8Frags& nxt) {
        constexpr int SLOT = decltype(slotc)::value;
        // this wave's loads younger than k-tile t + 1: k-tiles t + 2 .. t + NS - 2, four pieces each
        if (NS == 5 && t + 3 < T) wait_vm<8>();
        else if (t + 2 < T) wait_vm<4>();
        else wait_vm<0>();
        lds_barrier();          // k-tile t + 1 landed for every wave; every fragment read issued so far has returned
        if (t + NS - 1 < T) issue((SLOT + NS - 1) % NS);   // the slot of k-tile t - 1: its fragments were consumed by the previous step's MFMAs
        // hipcc's own LDS wait for `cur` lands here, in front of the reads issued below
#pragma unroll
        for (int i = 0; i < FM; ++i) asm volatile("" : "+v"(cur.a[i]));
#pragma unroll
        for (int j = 0; j < FN; ++j) asm volatile("" : "+v"(cur.b[j]));
        if (t + 1 < T) load_frags(nxt, std::integral_constant<int, (SLOT + 1) % NS>{});
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cur.b[j], cur.a[i], acc[i][j], 0, 0, 0);
        ++t;
    };
    // ring slot and fragment buffer are compile-time: the loop body is one trip round the ring (two for the odd ring: the fragment
    // buffers alternate); the leftover k-tiles continue at ring position 0
    constexpr int U = (NS % 2) ? 2 * NS : NS;
    auto run = [&](auto sc) {
        constexpr int S = decltype(sc)::value;
        if constexpr (S % 2 == 0) step(std::integral_constant<int, S % NS>{}, fa, fb);
        else step(std::integral_constant<int, S % NS>{}, fb, fa);
    };
#pragma unroll 1
    while (t + U <= T) static_for<U>(run);
    static_for<U - 1>([&](auto sc) { if (t < T) run(sc); });
    if (timing) T2 = cyc_now();
    wait_vm<0>();
    lds_barrier();                                       // every wave is done with the ring

    // ---- epilogue: rows through LDS, 16-byte accesses.  Finished where the MFMA leaves them (lane = row c16, 4 columns), output, residual
    // and mask move as 8-byte pieces 32 bytes apart -- 24 vector-memory instructions per lane, each touching a quarter of 16 cache lines:
    // 7.2k cycles of epilogue for 15k of k-loop on 12800 x 256 x 1024, 15k with residual + mask (tools/r3/gemm128_phases.py).  Instead
    // the two fragment rows a wave owns after the k-fold are laid out as [64 rows][128 columns] f32 in the idle ring (both k-half groups
    // at once), and every thread finishes 8 consecutive columns of a row: 16 lanes = one 256-byte row segment.  Residual / mask chunks
    // of all four passes are requested before the fold.
    const toist_epilogue& e = p.epi;
    const bf16_t* const resp = (const bf16_t*)e.res;
    const bf16_t* const auxp = (const bf16_t*)e.aux;
    bf16_t* const outp = (bf16_t*)p.c;
    const int act = e.act;
    const float alpha = e.alpha;
    const bool masked = act == TOIST_ACT_MASK_POS;
    constexpr int LDT = BN + 4;                          // f32 pitch of a band row (+4: conflict-free f32x4 writes)
    const int c8 = tid & 15, r_lo = tid >> 4;            // this thread's column chunk and its row inside a 32-row half band
    const int ncol = n0 + c8 * 8;
    const bool col_ok = ncol < N;                        // N % 8 == 0: a chunk is whole or absent
    float mul[8], add[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { mul[j] = alpha; add[j] = 0.f; }
    if (col_ok) {
        if (e.scale) {
            float t8[8];
            load_cols8(e.scale + ncol, 8, t8, 1.f);
#pragma unroll
            for (int j = 0; j < 8; ++j) mul[j] = alpha * t8[j];
        }
        if (e.shift) load_cols8(e.shift + ncol, 8, add, 0.f);
    }
    // pass (ii, q): band row 32 q + r_lo of the [64][128] layout = fragment row 2 q + ii of wave row (r_lo >> 4)
    int mrow[2][2];
    uint4 rres[2][2], raux[2][2];
#pragma unroll
    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int m = m0 + (r_lo >> 4) * WM + (q * 2 + ii) * 16 + (r_lo & 15);
            mrow[ii][q] = m;
            rres[ii][q] = raux[ii][q] = make_uint4(0u, 0u, 0u, 0u);
            if (col_ok && m < M) {
                if (resp) rres[ii][q] = *reinterpret_cast<const uint4*>(resp + (size_t)m * e.ldr + ncol);
                if (masked) raux[ii][q] = *reinterpret_cast<const uint4*>(auxp + (size_t)m * e.ldaux + ncol);
            }
        }
    // ---- fold the k-halves half and half: wave kh keeps fragment rows {2 kh, 2 kh + 1}, hands the other two to its partner (same wm, wn) ----
    float* const xch = reinterpret_cast<float*>(lds_raw);
#pragma unroll
    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
        for (int j = 0; j < FN; ++j)
            *reinterpret_cast<f32x4_t*>(xch + ((wave * 8 + ii * 4 + j) * 64 + lane) * 4) = kh == 0 ? acc[2 + ii][j] : acc[ii][j];
    lds_barrier();
    const int partner = wave ^ 4;
    f32x4_t fin[2][FN];
#pragma unroll
    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            const f32x4_t o = *reinterpret_cast<const f32x4_t*>(xch + ((partner * 8 + ii * 4 + j) * 64 + lane) * 4);
            fin[ii][j] = (kh == 0 ? acc[ii][j] : acc[2 + ii][j]) + o;
        }
    float* const band = xch + 8 * 8 * 64 * 4;            // behind the exchange area (64 KiB): [64][LDT] f32 = 33 KiB
#pragma unroll
    for (int ii = 0; ii < 2; ++ii) {
        if (ii) lds_barrier();                           // the previous pass is read
#pragma unroll
        for (int j = 0; j < FN; ++j) *reinterpret_cast<f32x4_t*>(band + (kh * 32 + wm * 16 + c16) * LDT + wn * WN + j * 16 + g * 4) = fin[ii][j];
        lds_barrier();
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int m = mrow[ii][q];
            if (!col_ok || m >= M) continue;
            const f32x4_t lo = *reinterpret_cast<const f32x4_t*>(band + (q * 32 + r_lo) * LDT + c8 * 8);
            const f32x4_t hi = *reinterpret_cast<const f32x4_t*>(band + (q * 32 + r_lo) * LDT + c8 * 8 + 4);
            float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = v[j] * mul[j] + add[j];
            if (resp) {
                float x[8];
                unpack8(rres[ii][q], x);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] += x[j];
            }
            if (act == TOIST_ACT_RELU) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
            } else if (masked) {
                float x[8];
                unpack8(raux[ii][q], x);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = x[j] > 0.f ? v[j] : 0.f;
            }
            *reinterpret_cast<uint4*>(outp + (size_t)m * p.ldc + ncol) = make_uint4(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7]));
        }
    }
    if (timing && lane == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        float* o = p.workspace + ((size_t)blockIdx.x * 8 + wave) * 8;
        o[4] = (float)T; o[5] = (float)(T1 - T0); o[6] = (float)(T2 - T1); o[7] = (float)(cyc_now() - T2);
    }
}

// gemm128w_kernel (tile code 137): the weight-gradient form of gemm128_kernel.  dW[co][(r, s, c)] = sum over pixels of dy[pixel][co] *
// x[pixel shifted by the tap][c]: BOTH operands are k-major (k = pixel), the reduction is 3 200 .. 51 200 deep and the output is a few
// dozen 128 x 128 tiles per convolution -- the identical residual blocks of a stage run as one grouped launch (blockIdx.y = problem x
// k-slice).  Same skeleton as above (4-slot DMA ring, k-halves on wave pairs, fragments one k-tile ahead); both fragment sets come
// out of LDS through ds_read_b64_tr_b16.  The 3 x 3 gather needs no table: a 128-column tile lies inside ONE tap (C % 128 == 0), so a
// lane's source address is its pixel's address plus a per-tile constant, and the only per-k-tile bookkeeping is the lane's running
// (y, x) -- advanced by 64 pixels per k-tile with two conditional subtractions -- against the tap's valid window (~20 VALU per k-tile
// against 16 MFMAs; the generic tile spends ~130 VALU + 60 SALU per 32 MFMAs on it).  Output: f32, alpha * rscale[m] (the folded
// FrozenBN scale), += into the gradient buffer, or raw k-slice partials into the caller's arena.
template <int BKD, int NS>   // NS = slots of the DMA ring: NS - 2 k-tiles (32 KiB each) in flight per CU behind the one being read
__global__ __launch_bounds__(512, 2) void gemm128w_kernel(const toist_gemm p) {
    constexpr int BM = G8_BM, BN = G8_BN, BK = G8_BK, WM = 64, WN = 64, FM = 4, FN = 4;
    constexpr bool GATHER = BKD == TOIST_B_CONVX;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kh = wave >> 2, wm = (wave >> 1) & 1, wn = wave & 1, g = lane >> 4, c16 = lane & 15;
    const int M = p.M, N = p.N, K = p.K;
    const int nt_n = (N + BN - 1) / BN, nt_m = (M + BM - 1) / BM;
    const int tiles = nt_m * nt_n;
    const int nz = p.batch * p.split_k;                 // (problem, k-slice) pairs
    // Workgroup -> (pair, tile).  With 128 x 128 tiles every operand k-tile is wanted by nt_n (A) or nt_m (B) workgroups: spread over
    // the XCDs, each of the 8 L2s pulls its own copy through the fabric (22 grouped layer-3 1x1 problems: 2.3 GB of operand tiles for
    // 0.7 GB of operands, 5-7 TB/s at the measured 320 us -- the launch was bound by that).  From 8 pairs on, a pair therefore lives on
    // ONE XCD: the dispatcher deals workgroups round-robin (id % 8 = XCD), so XCD x works through pairs x, x + 8, ... , all tiles of
    // a pair on consecutive ids of that XCD, walking K in step -- the pair's operands cross the fabric once.
    int z, tile_id;
    if (nz >= 8) {
        const int xcd = (int)(blockIdx.x & 7), sq = (int)(blockIdx.x >> 3);
        const int zi = sq / tiles;
        z = xcd + 8 * zi;
        tile_id = sq - zi * tiles;
        if (z >= nz) return;
    } else {
        z = blockIdx.y;
        tile_id = (int)(blockIdx.x & 7) * ((tiles + 7) >> 3) + (int)(blockIdx.x >> 3);
        if (tile_id >= tiles) return;
    }
    int tile_m, tile_n;
    tile_order(tile_id, nt_m, nt_n, tile_m, tile_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int bz = z / p.split_k, ksl = z - bz * p.split_k;
    const toist_operand ob = p.b;
    const bf16_t* a_ptr = (const bf16_t*)p.a.ptr;
    const bf16_t* b_ptr = (const bf16_t*)ob.ptr;
    long long coff = 0, rs_off = 0;
    if (p.group) {
        const toist_group gq = p.group[bz];
        a_ptr = (const bf16_t*)gq.a;
        b_ptr = (const bf16_t*)gq.b;
        coff = gq.c_off;
        rs_off = gq.rscale_off;
    }
    const int lda = p.a.ld, ldb = GATHER ? ob.SC : ob.ld;
    const int ktiles = K / BK;                          // K % 64 == 0
    const int kper = (ktiles + p.split_k - 1) / p.split_k;
    const int kt_beg = ksl * kper;
    const int T = ((kt_beg + kper < ktiles) ? kt_beg + kper : ktiles) - kt_beg;     // k-tiles of this slice
    if (T <= 0) return;
    // gather: this tile's tap and the window of pixel coordinates whose shifted source lies inside the plane
    int tap_dy = 0, tap_dx = 0, c0 = n0;
    if (GATHER) {
        const int tap = n0 / ob.SC;
        const int r = tap / ob.S, s_ = tap - r * ob.S;
        c0 = n0 - tap * ob.SC;
        tap_dy = r * ob.dil - ob.pad;
        tap_dx = s_ * ob.dil - ob.pad;
    }
    const i32x4_t rsA = make_rsrc(a_ptr);
    // strided gathers (conv2 / downsample of the first bottleneck of a stage): the source pixel of output pixel (img, y, x) is
    // (img, y * stride + dy, x * stride + dx) -- no longer the pixel's own address plus a constant, so the lane offset is rebuilt per k-tile
    const bool strided = GATHER && ob.stride != 1;
    const i32x4_t rsB = make_rsrc((GATHER && !strided) ? (const void*)(b_ptr + ((long long)tap_dy * ob.SW + tap_dx) * ob.SC) : (const void*)b_ptr);
    const unsigned lds0 = (unsigned)(size_t)lds_raw;

    // ---- DMA pieces: k-major tiles [64 k][128 m | n]; lane offsets in bytes, the k-tile in the scalar offset ----
    int va[2], vb[2], py[2], px[2], pimg[2];
    const int W = ob.PW, H = ob.PH;
    const int q64 = GATHER ? BK / W : 0, r64 = GATHER ? BK - q64 * W : 0;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int pch = (it * 8 + wave) * 64 + lane;
        const int krow = pch / (BM / 8), rc = swz_m<BM>(krow, pch % (BM / 8));
        const int mm = m0 + rc * 8, nn = n0 + rc * 8;
        va[it] = mm < M ? (mm + krow * lda) * 2 : OOB;
        vb[it] = nn < N ? ((GATHER ? c0 + rc * 8 : nn) + krow * ldb) * 2 : OOB;
        py[it] = px[it] = pimg[it] = 0;
        if (GATHER) {
            const int p0 = kt_beg * BK + krow;
            pimg[it] = p0 / (H * W);
            const int pix = p0 - pimg[it] * (H * W);
            py[it] = pix / W;
            px[it] = pix - py[it] * W;
            if (strided) vb[it] = nn < N ? (c0 + rc * 8) * 2 : OOB;        // the channel part only; the pixel part is rebuilt per k-tile
        }
    }
    int i_t = 0;                                        // k-tiles issued so far
    auto issue = [&](const int slot) {
        const unsigned da = lds0 + (unsigned)(slot * G8_STAGE_BYTES) + (unsigned)wave * 1024u, db = da + (unsigned)(G8_TILE * 2);
        const int kt = kt_beg + i_t;
        const int soa = kt * BK * lda * 2, sob = strided ? 0 : kt * BK * ldb * 2;
        int vbe[2] = {vb[0], vb[1]};
        if (GATHER) {
            if (strided) {
#pragma unroll
                for (int it = 0; it < 2; ++it) {        // selects, no nested branches (a branchy form of this was miscompiled for piece 0)
                    const int sy = py[it] * ob.stride + tap_dy, sx = px[it] * ob.stride + tap_dx;
                    const bool ok = (unsigned)sy < (unsigned)ob.SH && (unsigned)sx < (unsigned)ob.SW && vb[it] != OOB;
                    const int off = vb[it] + ((pimg[it] * ob.SH + sy) * ob.SW + sx) * ob.SC * 2;
                    vbe[it] = ok ? off : OOB;
                }
            } else {
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    const int sy = py[it] + tap_dy, sx = px[it] + tap_dx;
                    if (sy < 0 || sy >= H || sx < 0 || sx >= W) vbe[it] = OOB;
                }
            }
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                px[it] += r64;                          // the same k-row of the next k-tile: 64 pixels on
                py[it] += q64;
                if (px[it] >= W) { px[it] -= W; ++py[it]; }
                while (py[it] >= H) { py[it] -= H; ++pimg[it]; }
            }
        }
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %3, %7, %9 offen lds\n\t"
                     "s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %4, %7, %9 offen lds\n\t"
                     "s_mov_b32 m0, %11\n\ts_nop 0\n\tbuffer_load_dwordx4 %5, %8, %10 offen lds\n\t"
                     "s_mov_b32 m0, %12\n\ts_nop 0\n\tbuffer_load_dwordx4 %6, %8, %10 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "s"(da), "s"(da + 8192u), "v"(va[0]), "v"(va[1]), "v"(vbe[0]), "v"(vbe[1]), "s"(rsA), "s"(rsB), "s"(__builtin_amdgcn_readfirstlane(soa)),
                       "s"(__builtin_amdgcn_readfirstlane(sob)), "s"(db), "s"(db + 8192u)
                     : "memory");
        ++i_t;
    };
#pragma unroll
    for (int i = 0; i < NS - 1; ++i) issue(i);

    // ---- fragment addresses: [4 k][16 rows] blocks of the k-major tiles, transposed on the way out of LDS ----
    int a_base[FM][2], b_base[FN][2];
    {
        const int k = kh * 32 + 8 * g + (c16 >> 2), sub = (c16 & 1) * 4;
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const int rc = ((wm * WM + i * 16) >> 3) + ((c16 & 3) >> 1);
            a_base[i][0] = (k * BM + swz_m<BM>(k, rc) * 8 + sub) * 2;
            a_base[i][1] = ((k + 4) * BM + swz_m<BM>(k + 4, rc) * 8 + sub) * 2;
        }
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            const int rc = ((wn * WN + j * 16) >> 3) + ((c16 & 3) >> 1);
            b_base[j][0] = G8_TILE * 2 + (k * BN + swz_m<BN>(k, rc) * 8 + sub) * 2;
            b_base[j][1] = G8_TILE * 2 + ((k + 4) * BN + swz_m<BN>(k + 4, rc) * 8 + sub) * 2;
        }
    }
    auto load_frags = [&](G8Frags& f, auto slotc) {
        constexpr int SLOT = decltype(slotc)::value;
        typedef __attribute__((address_space(3))) s16x4_t* lds_v4;
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            union { struct { s16x4_t a, b; } h; bf16x8_t v; } u;
            u.h.a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(lds_raw + a_base[i][0] + SLOT * G8_STAGE_BYTES));
            u.h.b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(lds_raw + a_base[i][1] + SLOT * G8_STAGE_BYTES));
            f.a[i] = u.v;
        }
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            union { struct { s16x4_t a, b; } h; bf16x8_t v; } u;
            u.h.a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(lds_raw + b_base[j][0] + SLOT * G8_STAGE_BYTES));
            u.h.b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(lds_raw + b_base[j][1] + SLOT * G8_STAGE_BYTES));
            f.b[j] = u.v;
        }
    };

    f32x4_t acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    wait_vm<4 * (NS - 2)>();                            // k-tile 0 landed (the younger ones may fly; a slice has at least NS k-tiles)
    __builtin_amdgcn_s_barrier();
    G8Frags fa, fb;
    load_frags(fa, std::integral_constant<int, 0>{});
    int t = 0;
#ifdef TOIST_TUNING_KNOBS
    const int ABL = (p.flags >> 12) & 7;                // experiments: 1 = no MFMAs, 2 = no fragment reads, 4 = no DMA in the loop
#else
    constexpr int ABL = 0;
#endif
    auto step = [&](auto slotc, G8Frags& cur, G8Frags& nxt) {
        constexpr int SLOT = decltype(slotc)::value;
        // this wave's loads younger than k-tile t + 1: k-tiles t + 2 .. t + NS - 2, four pieces each
        if (NS == 5 && t + 3 < T) wait_vm<8>();
        else if (t + 2 < T) wait_vm<4>();
        else wait_vm<0>();
        lds_barrier();
        if (t + NS - 1 < T && !(ABL & 4)) issue((SLOT + NS - 1) % NS);
#pragma unroll
        for (int i = 0; i < FM; ++i) asm volatile("" : "+v"(cur.a[i]));
#pragma unroll
        for (int j = 0; j < FN; ++j) asm volatile("" : "+v"(cur.b[j]));
        if (t + 1 < T && !(ABL & 2)) load_frags(nxt, std::integral_constant<int, (SLOT + 1) % NS>{});
        __builtin_amdgcn_sched_barrier(0);
        if (!(ABL & 1)) {
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cur.b[j], cur.a[i], acc[i][j], 0, 0, 0);
        }
        ++t;
    };
    // ring slot and fragment buffer are compile-time: the loop body is one trip round the ring (two for an odd ring: the fragment
    // buffers alternate)
    constexpr int U = (NS % 2) ? 2 * NS : NS;
    auto run = [&](auto sc) {
        constexpr int S = decltype(sc)::value;
        if constexpr (S % 2 == 0) step(std::integral_constant<int, S % NS>{}, fa, fb);
        else step(std::integral_constant<int, S % NS>{}, fb, fa);
    };
#pragma unroll 1
    while (t + U <= T) static_for<U>(run);
    static_for<U - 1>([&](auto sc) { if (t < T) run(sc); });
    wait_vm<0>();
    lds_barrier();

    // ---- fold the k-halves half and half (as above), then f32 rows: lane = row c16 of a fragment, 4 consecutive columns (16 bytes) ----
    float* const xch = reinterpret_cast<float*>(lds_raw);
#pragma unroll
    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
        for (int j = 0; j < FN; ++j)
            *reinterpret_cast<f32x4_t*>(xch + ((wave * 8 + ii * 4 + j) * 64 + lane) * 4) = kh == 0 ? acc[2 + ii][j] : acc[ii][j];
    lds_barrier();
    const int partner = wave ^ 4;
    const toist_epilogue& e = p.epi;
    const bool partial = p.split_k > 1;
    float* const outp = partial ? p.workspace + ((size_t)bz * p.split_k + ksl) * M * N : (float*)p.c + coff;
    const int ldo = partial ? N : p.ldc;
    const float* const rsc = (!partial && e.rscale) ? e.rscale + rs_off : nullptr;
#pragma unroll
    for (int ii = 0; ii < 2; ++ii) {
        const int m = m0 + wm * WM + (kh * 2 + ii) * 16 + c16;
        const float rs = partial ? 1.f : (rsc && m < M ? e.alpha * rsc[m] : e.alpha);
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            const int n = n0 + wn * WN + j * 16 + g * 4;
            const f32x4_t o = *reinterpret_cast<const f32x4_t*>(xch + ((partner * 8 + ii * 4 + j) * 64 + lane) * 4);
            f32x4_t v = (kh == 0 ? acc[ii][j] : acc[2 + ii][j]) + o;
            if (m >= M || n >= N) continue;             // M, N % 8 == 0: the 4 columns are valid or absent together
            float* cp = outp + (size_t)m * ldo + n;
            if (!partial) {
                v *= rs;
                if (e.accumulate) v += *reinterpret_cast<const f32x4_t*>(cp);
            }
            *reinterpret_cast<f32x4_t*>(cp) = v;
        }
    }
}

// gemm256w_kernel (tile code 138): gemm128w_kernel with a 256 x 128 block tile.  The 128 x 128 weight-gradient kernel is bound by the
// L2 -> LDS stream (26 B/clk/CU with MFMAs and fragment reads switched off, profiles/r03_gemm128w_ablation.txt), so the next factor is flop
// per staged byte: 256 x 128 needs (256 + 128) / (256 * 128) = 3/4 of the bytes per flop, and layer 3's 256 output channels become ONE row of
// tiles (the grouped 3x3 launch: 18 instead of 36 workgroups per problem, 2 rounds per XCD instead of 4).  8 waves = 4 x 2 of 64 x 64, no k-split
// (so no fold in the epilogue); k-tiles of 32 pixels -- [32][256] + [32][128] bf16 = 24 KiB per ring slot, 6 slots, five k-tiles in flight --
// keep the 16 MFMAs per wave and barrier of the 128 x 128 kernel.
constexpr int G9_BM = 256, G9_BN = 128, G9_BK = 32, G9_NS = 6;
constexpr int G9_A_BYTES = G9_BM * G9_BK * 2, G9_STAGE_BYTES = (G9_BM + G9_BN) * G9_BK * 2;     // 16 KiB, 24 KiB
constexpr int G9_LDS = G9_NS * G9_STAGE_BYTES;                                                 // 144 KiB

template <int BKD>
__global__ __launch_bounds__(512, 2) void gemm256w_kernel(const toist_gemm p) {
    constexpr int BM = G9_BM, BN = G9_BN, BK = G9_BK, NS = G9_NS, WM = 64, WN = 64, FM = 4, FN = 4;
    constexpr bool GATHER = BKD == TOIST_B_CONVX;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, g = lane >> 4, c16 = lane & 15;
    const int M = p.M, N = p.N, K = p.K;
    const int nt_n = (N + BN - 1) / BN, nt_m = (M + BM - 1) / BM;
    const int tiles = nt_m * nt_n;
    const int nz = p.batch * p.split_k;
    int z, tile_id;                                     // (problem, k-slice) pairs pinned to XCDs from 8 pairs on (see gemm128w_kernel)
    if (nz >= 8) {
        const int xcd = (int)(blockIdx.x & 7), sq = (int)(blockIdx.x >> 3);
        const int zi = sq / tiles;
        z = xcd + 8 * zi;
        tile_id = sq - zi * tiles;
        if (z >= nz) return;
    } else {
        z = blockIdx.y;
        tile_id = (int)(blockIdx.x & 7) * ((tiles + 7) >> 3) + (int)(blockIdx.x >> 3);
        if (tile_id >= tiles) return;
    }
    const int tile_m = tile_id / nt_n, tile_n = tile_id - tile_m * nt_n;     // the N tiles of an M row run together: they share the A k-tiles
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int bz = z / p.split_k, ksl = z - bz * p.split_k;
    const toist_operand ob = p.b;
    const bf16_t* a_ptr = (const bf16_t*)p.a.ptr;
    const bf16_t* b_ptr = (const bf16_t*)ob.ptr;
    long long coff = 0, rs_off = 0;
    if (p.group) {
        const toist_group gq = p.group[bz];
        a_ptr = (const bf16_t*)gq.a;
        b_ptr = (const bf16_t*)gq.b;
        coff = gq.c_off;
        rs_off = gq.rscale_off;
    }
    const int lda = p.a.ld, ldb = GATHER ? ob.SC : ob.ld;
    const int k64 = K / 64;                             // the k-slices are cut in 64-row units (the host sizes its partials with that rule)
    const int kper = (k64 + p.split_k - 1) / p.split_k;
    const int kt_beg = 2 * ksl * kper;
    const int T = 2 * (((ksl + 1) * kper < k64 ? (ksl + 1) * kper : k64) - ksl * kper);     // 32-row k-tiles of this slice
    if (T <= 0) return;
    int tap_dy = 0, tap_dx = 0, c0 = n0;
    if (GATHER) {
        const int tap = n0 / ob.SC;
        const int r = tap / ob.S, s_ = tap - r * ob.S;
        c0 = n0 - tap * ob.SC;
        tap_dy = r * ob.dil - ob.pad;
        tap_dx = s_ * ob.dil - ob.pad;
    }
    const i32x4_t rsA = make_rsrc(a_ptr);
    const bool strided = GATHER && ob.stride != 1;       // see gemm128w_kernel
    const i32x4_t rsB = make_rsrc((GATHER && !strided) ? (const void*)(b_ptr + ((long long)tap_dy * ob.SW + tap_dx) * ob.SC) : (const void*)b_ptr);
    const unsigned lds0 = (unsigned)(size_t)lds_raw;

    // ---- DMA pieces per wave and k-tile: two of A ([32 k][256 m]), one of B ([32 k][128 n]) ----
    int va[2], vb, py = 0, px = 0, pimg = 0;
    const int W = ob.PW, H = ob.PH;
    const int q32 = GATHER ? BK / W : 0, r32 = GATHER ? BK - q32 * W : 0;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int pch = (it * 8 + wave) * 64 + lane;
        const int krow = pch / (BM / 8), rc = swz_m<BM>(krow, pch % (BM / 8));
        const int mm = m0 + rc * 8;
        va[it] = mm < M ? (mm + krow * lda) * 2 : OOB;
    }
    {
        const int pch = wave * 64 + lane;
        const int krow = pch / (BN / 8), rc = swz_m<BN>(krow, pch % (BN / 8));
        const int nn = n0 + rc * 8;
        vb = nn < N ? ((GATHER ? c0 + rc * 8 : nn) + krow * ldb) * 2 : OOB;
        if (GATHER) {
            const int p0 = kt_beg * BK + krow;
            pimg = p0 / (H * W);
            const int pix = p0 - pimg * (H * W);
            py = pix / W;
            px = pix - py * W;
            if (strided) vb = nn < N ? (c0 + rc * 8) * 2 : OOB;
        }
    }
    int i_t = 0;
    auto issue = [&](const int slot) {
        const unsigned da = lds0 + (unsigned)(slot * G9_STAGE_BYTES) + (unsigned)wave * 1024u, db = lds0 + (unsigned)(slot * G9_STAGE_BYTES + G9_A_BYTES) + (unsigned)wave * 1024u;
        const int kt = kt_beg + i_t;
        const int soa = kt * BK * lda * 2, sob = strided ? 0 : kt * BK * ldb * 2;
        int vbe = vb;
        if (GATHER) {
            if (strided) {
                const int sy = py * ob.stride + tap_dy, sx = px * ob.stride + tap_dx;
                const bool ok = (unsigned)sy < (unsigned)ob.SH && (unsigned)sx < (unsigned)ob.SW && vb != OOB;
                const int off = vb + ((pimg * ob.SH + sy) * ob.SW + sx) * ob.SC * 2;
                vbe = ok ? off : OOB;
            } else {
                const int sy = py + tap_dy, sx = px + tap_dx;
                if (sy < 0 || sy >= H || sx < 0 || sx >= W) vbe = OOB;
            }
            px += r32;
            py += q32;
            if (px >= W) { px -= W; ++py; }
            while (py >= H) { py -= H; ++pimg; }
        }
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %3, %6, %8 offen lds\n\t"
                     "s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %4, %6, %8 offen lds\n\t"
                     "s_mov_b32 m0, %10\n\ts_nop 0\n\tbuffer_load_dwordx4 %5, %7, %9 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "s"(da), "s"(da + 8192u), "v"(va[0]), "v"(va[1]), "v"(vbe), "s"(rsA), "s"(rsB), "s"(__builtin_amdgcn_readfirstlane(soa)),
                       "s"(__builtin_amdgcn_readfirstlane(sob)), "s"(db)
                     : "memory");
        ++i_t;
    };
#pragma unroll
    for (int i = 0; i < NS - 1; ++i) issue(i);         // a slice holds at least NS k-tiles (gemm256w_applies)

    int a_base[FM][2], b_base[FN][2];
    {
        const int k = 8 * g + (c16 >> 2), sub = (c16 & 1) * 4;
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const int rc = ((wm * WM + i * 16) >> 3) + ((c16 & 3) >> 1);
            a_base[i][0] = (k * BM + swz_m<BM>(k, rc) * 8 + sub) * 2;
            a_base[i][1] = ((k + 4) * BM + swz_m<BM>(k + 4, rc) * 8 + sub) * 2;
        }
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            const int rc = ((wn * WN + j * 16) >> 3) + ((c16 & 3) >> 1);
            b_base[j][0] = G9_A_BYTES + (k * BN + swz_m<BN>(k, rc) * 8 + sub) * 2;
            b_base[j][1] = G9_A_BYTES + ((k + 4) * BN + swz_m<BN>(k + 4, rc) * 8 + sub) * 2;
        }
    }
    auto load_frags = [&](G8Frags& f, auto slotc) {
        constexpr int SLOT = decltype(slotc)::value;
        typedef __attribute__((address_space(3))) s16x4_t* lds_v4;
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            union { struct { s16x4_t a, b; } h; bf16x8_t v; } u;
            u.h.a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(lds_raw + a_base[i][0] + SLOT * G9_STAGE_BYTES));
            u.h.b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(lds_raw + a_base[i][1] + SLOT * G9_STAGE_BYTES));
            f.a[i] = u.v;
        }
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            union { struct { s16x4_t a, b; } h; bf16x8_t v; } u;
            u.h.a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(lds_raw + b_base[j][0] + SLOT * G9_STAGE_BYTES));
            u.h.b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(lds_raw + b_base[j][1] + SLOT * G9_STAGE_BYTES));
            f.b[j] = u.v;
        }
    };

    f32x4_t acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    wait_vm<3 * (NS - 2)>();                            // k-tile 0 landed (three pieces per k-tile and wave)
    __builtin_amdgcn_s_barrier();
    G8Frags fa, fb;
    load_frags(fa, std::integral_constant<int, 0>{});
    int t = 0;
    auto step = [&](auto slotc, G8Frags& cur, G8Frags& nxt) {
        constexpr int SLOT = decltype(slotc)::value;
        // this wave's loads younger than k-tile t + 1: k-tiles t + 2 .. t + NS - 2
        if (t + 4 < T) wait_vm<9>();
        else if (t + 3 < T) wait_vm<6>();
        else if (t + 2 < T) wait_vm<3>();
        else wait_vm<0>();
        lds_barrier();
        if (t + NS - 1 < T) issue((SLOT + NS - 1) % NS);
#pragma unroll
        for (int i = 0; i < FM; ++i) asm volatile("" : "+v"(cur.a[i]));
#pragma unroll
        for (int j = 0; j < FN; ++j) asm volatile("" : "+v"(cur.b[j]));
        if (t + 1 < T) load_frags(nxt, std::integral_constant<int, (SLOT + 1) % NS>{});
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cur.b[j], cur.a[i], acc[i][j], 0, 0, 0);
        ++t;
    };
    auto run = [&](auto sc) {
        constexpr int S = decltype(sc)::value;
        if constexpr (S % 2 == 0) step(std::integral_constant<int, S % NS>{}, fa, fb);
        else step(std::integral_constant<int, S % NS>{}, fb, fa);
    };
#pragma unroll 1
    while (t + NS <= T) static_for<NS>(run);
    static_for<NS - 1>([&](auto sc) { if (t < T) run(sc); });

    // ---- f32 rows straight from the fragments: lane = row c16, 4 consecutive columns (16 bytes) ----
    const toist_epilogue& e = p.epi;
    const bool partial = p.split_k > 1;
    float* const outp = partial ? p.workspace + ((size_t)bz * p.split_k + ksl) * M * N : (float*)p.c + coff;
    const int ldo = partial ? N : p.ldc;
    const float* const rsc = (!partial && e.rscale) ? e.rscale + rs_off : nullptr;
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const int m = m0 + wm * WM + i * 16 + c16;
        const float rs = partial ? 1.f : (rsc && m < M ? e.alpha * rsc[m] : e.alpha);
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            const int n = n0 + wn * WN + j * 16 + g * 4;
            if (m >= M || n >= N) continue;
            f32x4_t v = acc[i][j];
            float* cp = outp + (size_t)m * ldo + n;
            if (!partial) {
                v *= rs;
                if (e.accumulate) v += *reinterpret_cast<const f32x4_t*>(cp);
            }
            *reinterpret_cast<f32x4_t*>(cp) = v;
        }
    }
}

#endif  // GEMM_UNIT (128x128 kernels)

static bool aligned16(const void* p) { return (((size_t)p) & 15) == 0; }

// what epilogue_lean (and the panel kernel's epilogue) covers
static bool lean_epilogue_ok(const toist_gemm& d) {
    // TOIST_LEAN_EPILOGUE: 0 = never, 1 = only scale / shift / residual / ReLU / mask, 2 (default) = also dropout, GELU, aux-based gradients, row map
    static const int level = (int)tuning_knob("TOIST_LEAN_EPILOGUE", 2);
    const toist_epilogue& e = d.epi;
    if (level == 1 && (e.drop_where || e.cmap || (e.act != TOIST_ACT_NONE && e.act != TOIST_ACT_RELU && e.act != TOIST_ACT_MASK_POS))) return false;
    if (level <= 0 || d.split_k > 1 || d.group != nullptr || d.a_colsum != nullptr) return false;
    if (e.out_f32 || e.accumulate || e.rscale || e.pre_out || e.res_div > 0) return false;
    if (e.act == TOIST_ACT_SIGMOID) return false;
    if ((d.N % 8) != 0 || (d.ldc % 8) != 0 || !aligned16(d.c) || (d.cs_outer % 8) != 0 || (d.cs_inner % 8) != 0) return false;
    if (e.res && ((e.ldr % 8) != 0 || !aligned16(e.res))) return false;
    if (e.act >= TOIST_ACT_MASK_POS && (e.aux == nullptr || (e.ldaux % 8) != 0 || !aligned16(e.aux))) return false;
    if ((e.scale && (((size_t)e.scale) & 15)) || (e.shift && (((size_t)e.shift) & 15))) return false;
    return true;
}

// ---- short reductions (K <= 256): one resident weight panel per workgroup ------------------------------------------------
// The 1x1 convolutions of ResNet layers 1-3 and their data gradients are GEMMs with K = 64 .. 256 and tens of thousands of rows
// (12800 x 1024 x 256 at batch 8: 45 launches per step).  Through the generic tiles every 64x64 output tile stages 64 KB of
// operands for 4 k-steps of MFMAs and then a 3-stream epilogue (residual, ReLU mask, store): the launch is bound by the LDS-DMA
// rate in its reduction loops, by HBM in its epilogues, and the two phases did not overlap (tools/dbg/gemm_attr.py: 26.7 us =
// 6.9 dispatch + 8.9 reduction + 10.9 epilogue).  Here a workgroup keeps ONE 64-column panel of B for its whole life -- as MFMA
// fragments in registers (64 VGPRs for K = 256) -- and walks down the rows: per output tile only the 64 x K block of A travels
// (half the staged bytes), whole-K at once into one of two 32 KB LDS slots, and the block of the NEXT tile is in flight while
// this tile's epilogue reads its residual / mask rows and stores.  Workgroups are dealt to XCDs (blockIdx % 8) so that the
// N / 64 workgroups sharing an A block sit on one L2; 64 KB of LDS = two workgroups per CU.
template <int BKD, int ACT>
__global__ __launch_bounds__(256, 2) void panel_kernel(const toist_gemm p) {
    constexpr int BM = 64, BN = 64, BK = 64, WM = 32, WN = 32, FM = 2, FN = 2, KT = 4, KS = 8;
    constexpr int SUB = BM * BK;                 // one k-tile of an A block (8 KiB)
    constexpr int SLOT = SUB * KT;               // whole-K A block (32 KiB)
    constexpr int LDT = BN + 4;                  // f32 band pitch
    __shared__ __attribute__((aligned(16))) bf16_t smem[2 * SLOT];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, g = lane >> 4, c16 = lane & 15;
    const int M = p.M, N = p.N, K = p.K;
    const int nt_m = (M + BM - 1) / BM, nt_n = (N + BN - 1) / BN;
    const int kt = (K + BK - 1) / BK, nks = (K + 31) / 32;
    // blockIdx = 8 * slot + xcd: XCD x owns the contiguous row tiles [x * m_per, ...); inside it workgroup `slot` = (row group, column panel)
    const int xcd = (int)blockIdx.x & 7, slot_id = (int)blockIdx.x >> 3;
    const int groups = ((int)gridDim.x >> 3) / nt_n;
    const int tile_n = slot_id % nt_n, rr = slot_id / nt_n;
    const int m_per = (nt_m + 7) >> 3;
    const int m_beg = xcd * m_per, m_end = (m_beg + m_per < nt_m) ? m_beg + m_per : nt_m;
    if (rr >= groups || m_beg + rr >= m_end) return;
    const int n0 = tile_n * BN;
    const i32x4_t rsA = make_rsrc(p.a.ptr), rsB = make_rsrc(p.b.ptr);
    const int lda = p.a.ld, ldb = p.b.ld;
    const unsigned lds0 = (unsigned)(size_t)smem;

    // ---- the B panel as MFMA fragments: bq[ks][j] = 8 consecutive k (32 ks + 8 g ..) of column n0 + wn*32 + 16 j + c16 ----
    bf16x8_t bq[KS][FN];
    if (BKD == TOIST_B_ROWK) {
        const bf16_t* B = (const bf16_t*)p.b.ptr;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                const int n = n0 + wn * WN + j * 16 + c16, k = ks * 32 + g * 8;
                const bf16x8_t zero = {0, 0, 0, 0, 0, 0, 0, 0};
                bq[ks][j] = (n < N && k < K) ? *reinterpret_cast<const bf16x8_t*>(B + (size_t)n * ldb + k) : zero;
            }
    } else {
        // k-major weights [K][N]: staged once through LDS in the k-major tile layout, transposed into fragments by ds_read_b64_tr_b16
#pragma unroll
        for (int t = 0; t < KT; ++t)
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int pch = (it * 4 + wave) * 64 + lane;
                const int krow = pch / (BN / 8), rc = swz_m<BN>(krow, pch % (BN / 8));
                const int nn = n0 + rc * 8, k = t * BK + krow;
                if (t < kt) dma16(lds0 + (unsigned)(t * SUB * 2) + (unsigned)(it * 4 + wave) * 1024u, rsB, nn + k * ldb, nn < N && k < K);
            }
        wait_vm<0>();
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                const bf16x8_t zero = {0, 0, 0, 0, 0, 0, 0, 0};
                bq[ks][j] = (ks < nks) ? fragment<true, BN, BK>(smem + (ks >> 1) * SUB, wn * WN + j * 16, ks & 1, g, c16) : zero;
            }
        __syncthreads();
    }

    // ---- A blocks: k-tile t of a block is a [64][64] k-contiguous tile (2 pieces per thread) ----
    int a_off[2];                                 // row * lda + swizzled k-chunk of this thread's two pieces
    int a_row[2], a_col[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int pch = (it * 4 + wave) * 64 + lane;
        a_row[it] = pch / (BK / 8);
        a_col[it] = swz_k<BK>(a_row[it], pch % (BK / 8)) * 8;
        a_off[it] = a_row[it] * lda + a_col[it];
    }
    auto issue_a = [&](int slot, int tile_m) {
        const int m0 = tile_m * BM;
#pragma unroll
        for (int t = 0; t < KT; ++t)
#pragma unroll
            for (int it = 0; it < 2; ++it)
                if (t < kt) dma16(lds0 + (unsigned)((slot * SLOT + t * SUB) * 2) + (unsigned)(it * 4 + wave) * 1024u, rsA, m0 * lda + a_off[it] + t * BK,
                                  m0 + a_row[it] < M && t * BK + a_col[it] < K);
    };

    // ---- lean epilogue (panel_applies admits only what it covers): bf16 rows of 8 columns, 16-byte accesses, optional per-column scale /
    // shift, optional residual, ACT in {none, ReLU, mask by aux > 0}.  A thread finishes chunk c8 of band row `brow` of both bands; the
    // residual / mask chunks of a tile are requested one tile ahead. ----
    const int c8 = tid & 7, brow = tid >> 3;
    const int ncol = n0 + c8 * 8;
    const bool col_ok = ncol < N;                 // N % 8 == 0: a chunk is whole or absent
    const int rloc = (brow >> 4) * WM + (brow & 15);            // + 16 i: tile row of this thread in band i
    const bf16_t* const resp = (const bf16_t*)p.epi.res;
    const bf16_t* const auxp = (const bf16_t*)p.epi.aux;
    bf16_t* const outp = (bf16_t*)p.c;
    const int ldc = p.ldc, ldr = p.epi.ldr, ldaux = p.epi.ldaux;
    const float alpha = p.epi.alpha;
    const int drop = p.epi.drop_where;            // nn.Linear + dropout of the transformer layers (same element index and hash as epilogue_row8)
    const unsigned long long dseed = drop ? p.epi.drop_seed + (p.epi.drop_seed_dev ? *p.epi.drop_seed_dev : 0ull) : 0ull;
    const unsigned dth = (unsigned)(p.epi.drop_p * 4294967296.0);
    const float dsc = 1.f / (1.f - p.epi.drop_p);
    float csc[8], csh[8];
    const bool has_scale = p.epi.scale != nullptr, has_shift = p.epi.shift != nullptr;
#pragma unroll
    for (int j = 0; j < 8; ++j) { csc[j] = 1.f; csh[j] = 0.f; }
    if (col_ok) {
        if (has_scale) load_cols8(p.epi.scale + ncol, 8, csc, 1.f);
        if (has_shift) load_cols8(p.epi.shift + ncol, 8, csh, 0.f);
    }
    struct Rows { uint4 res[FM], aux[FM]; };
    auto request = [&](const int tile_m, Rows& q) {
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const int m = tile_m * BM + rloc + 16 * i;
            if (col_ok && m < M) {
                if (resp) q.res[i] = *reinterpret_cast<const uint4*>(resp + (size_t)((unsigned)(m * ldr + ncol)));
                if (ACT == TOIST_ACT_MASK_POS) q.aux[i] = *reinterpret_cast<const uint4*>(auxp + (size_t)((unsigned)(m * ldaux + ncol)));
            }
        }
    };

    Rows q0, q1;
    int tm = m_beg + rr;
    issue_a(0, tm);
    request(tm, q0);
    // one tile: block in `slot`, rows in `cur`; the next tile's block / rows go to the other slot / `nxt` (two copies of the body, so
    // that the row state stays in registers)
    auto tile = [&](const int slot, Rows& cur, Rows& nxt) {
        wait_vm<0>();                     // this tile's block and residual / mask rows landed (and the previous stores drained)
        __builtin_amdgcn_s_barrier();     // ... for every wave; everyone is done with the other slot (previous tile's band)
        if (tm + groups < m_end) {        // both fly during the MFMAs and the epilogue below
            issue_a(slot ^ 1, tm + groups);
            request(tm + groups, nxt);
        }
        f32x4_t acc[FM][FN];
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        const bf16_t* sA = smem + slot * SLOT;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (ks < nks) {
                bf16x8_t af[FM];
#pragma unroll
                for (int i = 0; i < FM; ++i) af[i] = fragment<false, BM, BK>(sA + (ks >> 1) * SUB, wm * WM + i * 16, ks & 1, g, c16);
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int j = 0; j < FN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bq[ks][j], af[i], acc[i][j], 0, 0, 0);
            }
        }
        // bands of 32 rows through the (consumed) A slot: fragment row i of both wave rows, then one 8-column chunk per thread
        float* const band = reinterpret_cast<float*>(smem + slot * SLOT);
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            lds_barrier();                // the MFMA reads of this slot / the previous band's reads are done
#pragma unroll
            for (int j = 0; j < FN; ++j) *reinterpret_cast<f32x4_t*>(band + (wm * 16 + c16) * LDT + wn * WN + j * 16 + g * 4) = acc[i][j];
            lds_barrier();
            const int m = tm * BM + rloc + 16 * i;
            if (col_ok && m < M) {
                const f32x4_t lo = *reinterpret_cast<const f32x4_t*>(band + brow * LDT + c8 * 8);
                const f32x4_t hi = *reinterpret_cast<const f32x4_t*>(band + brow * LDT + c8 * 8 + 4);
                float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = v[j] * (alpha * csc[j]) + csh[j];
                const unsigned long long didx = (unsigned long long)m * N + ncol;
                if (drop == 1) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = dropout_keep(dseed, didx + j, dth) ? v[j] * dsc : 0.f;
                }
                if (resp) {
                    float x[8];
                    unpack8(cur.res[i], x);
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] += x[j];
                }
                if (ACT == TOIST_ACT_RELU) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
                }
                if (ACT == TOIST_ACT_MASK_POS) {
                    float x[8];
                    unpack8(cur.aux[i], x);
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = x[j] > 0.f ? v[j] : 0.f;
                }
                if (drop == 2) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = dropout_keep(dseed, didx + j, dth) ? v[j] * dsc : 0.f;
                }
                *reinterpret_cast<uint4*>(outp + (size_t)((unsigned)(m * ldc + ncol))) =
                    make_uint4(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7]));
            }
        }
    };
    while (tm < m_end) {
        tile(0, q0, q1);
        tm += groups;
        if (tm >= m_end) break;
        tile(1, q1, q0);
        tm += groups;
    }
}

// ---- panel kernel, second generation (round 3) ------------------------------------------------------------------------------
// What bounded panel_kernel above (profiles/r02: 2.5-2.8 TB/s of algorithmic bytes = 31-36 % of the HBM peak, traffic 1.06 x): every
// tile opened with `s_waitcnt vmcnt(0)`, which on gfx9 also drains the STORES the wave issued a few hundred cycles earlier (stores and
// loads share the counter) -- a full write round trip per tile -- and the residual / mask rows were compiler-visible register loads
// requested ONE tile ahead, so a workgroup had 8-16 KB of HBM reads in flight against the ~49 KB per CU that 5 TB/s x 2.5 us asks for.
// Here nothing the compiler can see touches memory inside the tile loop:
//   * the residual and mask tiles of a row block travel by LDS-DMA into the same ring stage as its A block (no VGPRs held by loads in
//     flight, any prefetch distance);
//   * the output rows leave through `buffer_store_dwordx4` in inline asm (out-of-range lanes use an offset beyond num_records: the
//     instruction is issued by every wave of every tile, which makes the per-tile operation count exact);
//   * the tile-top wait is `vmcnt(n)` with n = everything this wave issued AFTER the loads of the tile it is about to read
//     (memory operations of one wave retire from the counter in issue order): the stores of the last D tiles and the loads of the next
//     D - 1 stay in flight across the barrier.
// ns ring stages (prefetch distance D = ns - 1 tiles) of [A block BM x K | residual BM x 64 | mask BM x 64]; BM = 64 or 32 rows.
__device__ __forceinline__ void wait_vm_n(const int n) {   // wave-uniform n; a smaller immediate (waiting for more) is always safe
#define TOIST_W(i) case i: wait_vm<i>(); break;
    switch (n) {
        TOIST_W(0) TOIST_W(1) TOIST_W(2) TOIST_W(3) TOIST_W(4) TOIST_W(5) TOIST_W(6) TOIST_W(7) TOIST_W(8) TOIST_W(9) TOIST_W(10) TOIST_W(11)
        TOIST_W(12) TOIST_W(13) TOIST_W(14) TOIST_W(15) TOIST_W(16) TOIST_W(17) TOIST_W(18) TOIST_W(19) TOIST_W(20) TOIST_W(21) TOIST_W(22)
        TOIST_W(23) TOIST_W(24) TOIST_W(25) TOIST_W(26) TOIST_W(27) TOIST_W(28) TOIST_W(29) TOIST_W(30) TOIST_W(31) TOIST_W(32) TOIST_W(33)
        TOIST_W(34) TOIST_W(35) TOIST_W(36) TOIST_W(37) TOIST_W(38) TOIST_W(39) TOIST_W(40) TOIST_W(41) TOIST_W(42) TOIST_W(43) TOIST_W(44)
        default: wait_vm<45>(); break;
    }
#undef TOIST_W
}

// 16 bytes of every lane to (descriptor base + byte offset); an offset >= num_records drops the lane's store.  The trailing s_nop keeps
// hipcc's next instruction off the data registers until the store has read them (cdna_hip_programming.md 5.7).
__device__ __forceinline__ void store16_asm(const i32x4_t& r, const int byte_off, const u32x4_t& v) {
    asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen\n\ts_nop 1" ::"v"(v), "v"(byte_off), "s"(r) : "memory");
}

constexpr int P2_HEAD = 0;     // ring stages start here (the epilogue needs no LDS band: fragments are finished where the MFMA left them)

// 8 bytes of every lane (see store16_asm)
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;
__device__ __forceinline__ void store8_asm(const i32x4_t& r, const int byte_off, const u32x2_t& v) {
    asm volatile("buffer_store_dwordx2 %0, %1, %2, 0 offen\n\ts_nop 1" ::"v"(v), "v"(byte_off), "s"(r) : "memory");
}

// PROF (experiments, flags bit 11 + a workspace): wave 0 of every workgroup adds up the shader cycles of its four tile phases
template <int BKD, int ACT, int BM, bool PROF = false>
__global__ __launch_bounds__(256, 2) void panel2_kernel(const toist_gemm p, const int ns) {
    constexpr int BN = 64, BK = 64, WM = BM / 2, WN = 32, FM = WM / 16, FN = 2, KS = 8;
    constexpr int SUB = BM * BK;                  // one k-tile of an A block (elements)
    constexpr int PW = BM / 32;                   // 1 KiB pieces per wave: per A k-tile, per residual tile, per mask tile
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16_t* const smem = reinterpret_cast<bf16_t*>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, g = lane >> 4, c16 = lane & 15;
    const int M = p.M, N = p.N, K = p.K;
    const int nt_m = (M + BM - 1) / BM, nt_n = (N + BN - 1) / BN;
    const int kt = (K + BK - 1) / BK, nks = (K + 31) / 32;
    const int xcd = (int)blockIdx.x & 7, slot_id = (int)blockIdx.x >> 3;
    const int groups = ((int)gridDim.x >> 3) / nt_n;
    const int tile_n = slot_id % nt_n, rr = slot_id / nt_n;
    const int m_per = (nt_m + 7) >> 3;
    const int m_beg = xcd * m_per, m_end = (m_beg + m_per < nt_m) ? m_beg + m_per : nt_m;
    if (rr >= groups || m_beg + rr >= m_end) return;
    const int T = (m_end - (m_beg + rr) + groups - 1) / groups;      // tiles of this workgroup: m_beg + rr + j * groups
    const int n0 = tile_n * BN;
    // Column order of the MFMA fragments (round 5): row q of fragment j of the B operand is column 8 (q >> 2) + 4 j + (q & 3) of the wave's 32, so
    // that the 2 x 4 values a lane holds after the MFMA -- rows 4 g .. 4 g + 3 of both fragments -- are the EIGHT consecutive columns 8 g .. 8 g + 7
    // of one output row: one 16-byte residual / mask read and one 16-byte store per lane and row instead of two 8-byte ones.  Sums are unchanged.
    auto pcol = [](const int j, const int q) { return 8 * (q >> 2) + 4 * j + (q & 3); };
    const bf16_t* const resp = (const bf16_t*)p.epi.res;
    const bf16_t* const auxp = (const bf16_t*)p.epi.aux;
    const bool has_res = resp != nullptr;
    constexpr bool has_aux = ACT == TOIST_ACT_MASK_POS;
    const i32x4_t rsA = make_rsrc(p.a.ptr), rsB = make_rsrc(p.b.ptr), rsC = make_rsrc(p.c);
    const i32x4_t rsR = make_rsrc(has_res ? (const void*)resp : p.a.ptr), rsX = make_rsrc(has_aux ? (const void*)auxp : p.a.ptr);
    const int lda = p.a.ld, ldb = p.b.ld;
    const unsigned lds0 = (unsigned)(size_t)smem;
    // stage layout (elements): [kt k-tiles of A][residual tile][mask tile]
    const int res_off = kt * SUB, aux_off = res_off + (has_res ? BM * BN : 0), stage = aux_off + (has_aux ? BM * BN : 0);
    bf16_t* const ring = smem + P2_HEAD / 2;
    const unsigned ring0 = lds0 + P2_HEAD;

    // ---- the B panel as MFMA fragments (as in panel_kernel) ----
    bf16x8_t bq[KS][FN];
    if (BKD == TOIST_B_ROWK) {
        const bf16_t* B = (const bf16_t*)p.b.ptr;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                const int n = n0 + wn * WN + pcol(j, c16), k = ks * 32 + g * 8;
                const bf16x8_t zero = {0, 0, 0, 0, 0, 0, 0, 0};
                bq[ks][j] = (n < N && k < K) ? *reinterpret_cast<const bf16x8_t*>(B + (size_t)n * ldb + k) : zero;
            }
    } else {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int pch = (it * 4 + wave) * 64 + lane;
                const int krow = pch / (BN / 8), rc = swz_m<BN>(krow, pch % (BN / 8));
                const int nn = n0 + rc * 8, k = t * BK + krow;
                if (t < kt) dma16(lds0 + (unsigned)(t * 64 * BK * 2) + (unsigned)(it * 4 + wave) * 1024u, rsB, nn + k * ldb, nn < N && k < K);
            }
        wait_vm<0>();
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                const bf16x8_t zero = {0, 0, 0, 0, 0, 0, 0, 0};
                bq[ks][j] = (ks < nks) ? fragment_perm8<BN, BK>(smem + (ks >> 1) * 64 * BK, wn * WN, j, ks & 1, g, c16) : zero;
            }
        __syncthreads();
    }

    // ---- per-lane invariants of the DMA pieces ----
    int a_off[PW], a_row[PW], a_col[PW];          // A k-tile: [BM][64] k-contiguous, swizzled
    int r_off[PW], x_off[PW], r_row[PW];          // residual / mask tile: [BM][64] row-major, lane-linear
    bool r_colok[PW];
#pragma unroll
    for (int it = 0; it < PW; ++it) {
        const int pch = (it * 4 + wave) * 64 + lane;
        a_row[it] = pch / (BK / 8);
        a_col[it] = swz_k<BK>(a_row[it], pch % (BK / 8)) * 8;
        a_off[it] = a_row[it] * lda + a_col[it];
        r_row[it] = pch >> 3;
        const int cc = n0 + ((pch & 7) ^ ((r_row[it] >> 1) & 7)) * 8;       // 16-byte chunk c of row r sits in slot c ^ ((r >> 1) & 7): conflict-free 8-byte fragment reads
        r_colok[it] = cc < N;
        r_off[it] = r_row[it] * p.epi.ldr + cc;
        x_off[it] = r_row[it] * p.epi.ldaux + cc;
    }
    const int P = PW * (kt + (has_res ? 1 : 0) + (has_aux ? 1 : 0));    // DMA instructions per wave per tile
    constexpr int S = FM;                                               // store instructions per wave per tile
    // Lane offsets of a FULL tile (every row < M): loop invariants in bytes, out-of-range where the K tail / the column tail says so; the
    // tile's row offset m0 * ld travels in the buffer instruction's scalar offset -- no vector instruction per piece.  (The DMA issue
    // of a tile used to be ~90 instructions: address adds, predicates and an M0 save / restore per piece.)
    int va[4][PW], vr[PW], vx[PW];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int it = 0; it < PW; ++it) va[t][it] = (t < kt && t * BK + a_col[it] < K) ? (a_off[it] + t * BK) * 2 : OOB;
#pragma unroll
    for (int it = 0; it < PW; ++it) {
        vr[it] = r_colok[it] ? r_off[it] * 2 : OOB;
        vx[it] = r_colok[it] ? x_off[it] * 2 : OOB;
    }
    auto issue = [&](const int slot, const int tile_m) {
        const int m0 = tile_m * BM;
        const unsigned sb = ring0 + (unsigned)(slot * stage) * 2u + (unsigned)wave * 1024u;
        if (m0 + BM <= M) {
            const int so = m0 * lda * 2;
#pragma unroll
            for (int t = 0; t < 4; ++t)
                if (t < kt) {
                    if constexpr (PW == 2) dma16s2(sb + (unsigned)(t * SUB * 2), rsA, va[t][0], va[t][1], so);
                    else dma16s(sb + (unsigned)(t * SUB * 2), rsA, va[t][0], so);
                }
            if (has_res) {
                if constexpr (PW == 2) dma16s2(sb + (unsigned)(res_off * 2), rsR, vr[0], vr[1], m0 * p.epi.ldr * 2);
                else dma16s(sb + (unsigned)(res_off * 2), rsR, vr[0], m0 * p.epi.ldr * 2);
            }
            if (has_aux) {
                if constexpr (PW == 2) dma16s2(sb + (unsigned)(aux_off * 2), rsX, vx[0], vx[1], m0 * p.epi.ldaux * 2);
                else dma16s(sb + (unsigned)(aux_off * 2), rsX, vx[0], m0 * p.epi.ldaux * 2);
            }
            return;
        }
        // the last, partial tile of the matrix: per-lane row predicates (same number of instructions per wave)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int it = 0; it < PW; ++it)
                if (t < kt) dma16(sb + (unsigned)(t * SUB * 2) + (unsigned)(it * 4) * 1024u, rsA, m0 * lda + a_off[it] + t * BK,
                                  m0 + a_row[it] < M && t * BK + a_col[it] < K);
        if (has_res) {
#pragma unroll
            for (int it = 0; it < PW; ++it)
                dma16(sb + (unsigned)(res_off * 2) + (unsigned)(it * 4) * 1024u, rsR, m0 * p.epi.ldr + r_off[it], m0 + r_row[it] < M && r_colok[it]);
        }
        if (has_aux) {
#pragma unroll
            for (int it = 0; it < PW; ++it)
                dma16(sb + (unsigned)(aux_off * 2) + (unsigned)(it * 4) * 1024u, rsX, m0 * p.epi.ldaux + x_off[it], m0 + r_row[it] < M && r_colok[it]);
        }
    };

    // ---- epilogue invariants: a lane finishes, for each of its FM x FN fragments, the 4 consecutive columns of ONE row the MFMA left it
    // (the arithmetic of panel_kernel / epilogue_lean, element for element): no LDS band, no barrier, 8-byte accesses ----
    const int ldc = p.ldc;
    const float alpha = p.epi.alpha;
    const int drop = p.epi.drop_where;
    const unsigned long long dseed = drop ? p.epi.drop_seed + (p.epi.drop_seed_dev ? *p.epi.drop_seed_dev : 0ull) : 0ull;
    const unsigned dth = (unsigned)(p.epi.drop_p * 4294967296.0);
    const float dsc = 1.f / (1.f - p.epi.drop_p);
    static_assert(FN == 2, "the fragment column order above pairs two fragments");
    const int ncol = n0 + wn * WN + g * 8;              // first of this lane's 8 columns (fragment 0: +0..3, fragment 1: +4..7)
    const bool col_ok = ncol < N;                       // N % 8 == 0: the 8 columns are valid or absent together
    float csc[FN][4], csh[FN][4];
#pragma unroll
    for (int jj = 0; jj < FN; ++jj) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { csc[jj][e] = alpha; csh[jj][e] = 0.f; }
        if (col_ok) {
            if (p.epi.scale != nullptr) {
                const float4 t4 = *reinterpret_cast<const float4*>(p.epi.scale + ncol + jj * 4);
                csc[jj][0] = alpha * t4.x; csc[jj][1] = alpha * t4.y; csc[jj][2] = alpha * t4.z; csc[jj][3] = alpha * t4.w;
            }
            if (p.epi.shift != nullptr) {
                const float4 t4 = *reinterpret_cast<const float4*>(p.epi.shift + ncol + jj * 4);
                csh[jj][0] = t4.x; csh[jj][1] = t4.y; csh[jj][2] = t4.z; csh[jj][3] = t4.w;
            }
        }
    }
    // residual / mask row piece of this lane inside a stage's [BM][64] tile: row wm*WM + 16 i + c16, the 16-byte chunk wn*4 + g (stored in slot
    // chunk ^ ((row >> 1) & 7): the 16 rows of a 16-lane read pass cover all 64 banks once)
    int rx_off[FM];
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const int r = wm * WM + i * 16 + c16, c = wn * 4 + g;
        rx_off[i] = r * BN + ((c ^ ((r >> 1) & 7)) * 8);
    }
    // The prologue's own (compiler-visible) loads end here: naming their registers as asm operands makes hipcc place ITS wait for them in
    // front of this statement instead of at their first use inside the tile loop, where it would be a vmcnt(0) per tile.  From here on
    // the wave's counter holds only what `issue` and the stores put there.
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int jj = 0; jj < FN; ++jj) asm volatile("" : "+v"(bq[ks][jj]));
#pragma unroll
    for (int jj = 0; jj < FN; ++jj)
#pragma unroll
        for (int e = 0; e < 4; ++e) asm volatile("" : "+v"(csc[jj][e]), "+v"(csh[jj][e]));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    const int D = ns - 1;
#pragma unroll 1
    for (int j = 0; j < D && j < T; ++j) issue(j, m_beg + rr + j * groups);
    int slot = 0;
    unsigned long long pc[4] = {0, 0, 0, 0}, t0 = 0, t1 = 0;
    if (PROF) t0 = cyc_now();
#pragma unroll 1
    for (int j = 0; j < T; ++j) {
        const int tm = m_beg + rr + j * groups;
        // operations this wave issued after the loads of tile j: the stores of the last min(j, D) tiles, the loads of the next min(D - 1, T - 1 - j)
        const int st_young = (j < D ? j : D) * S, ld_young = (T - 1 - j < D - 1 ? T - 1 - j : D - 1) * P;
        wait_vm_n(st_young + ld_young);
        asm volatile("s_barrier" ::: "memory");           // ... landed for every wave; everyone is done with the stage tile j - 1 was read from
        if (PROF) { t1 = cyc_now(); pc[0] += t1 - t0; t0 = t1; }
        if (j + D < T) {
            int s2 = slot + D;
            if (s2 >= ns) s2 -= ns;
            issue(s2, tm + D * groups);
        }
        if (PROF) { t1 = cyc_now(); pc[1] += t1 - t0; t0 = t1; }
        f32x4_t acc[FM][FN];
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int jj = 0; jj < FN; ++jj) acc[i][jj] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        const bf16_t* sA = ring + slot * stage;
        // All A fragments of the tile are read before the first MFMA (round 5): with the k-step count a run-time value the compiler emitted
        // read -> wait -> two MFMAs per step, eight LDS latencies in a row (the "mfma" phase of profiles/r03_panel2_phase_cycles.txt: 1050
        // cycles for 16 MFMAs).  The reduction depths of the hot path (K = 256 / 128 / 64) get a straight-line body each; same MFMA order.
        auto mma_all = [&](auto nk_tag) {
            constexpr int NK = decltype(nk_tag)::value;
            bf16x8_t af[NK][FM];
#pragma unroll
            for (int ks = 0; ks < NK; ++ks)
#pragma unroll
                for (int i = 0; i < FM; ++i) af[ks][i] = fragment<false, BM, BK>(sA + (ks >> 1) * SUB, wm * WM + i * 16, ks & 1, g, c16);
            __builtin_amdgcn_sched_barrier(0);      // (the scheduler otherwise sinks every read to its MFMA again)
#pragma unroll
            for (int ks = 0; ks < NK; ++ks)
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int jj = 0; jj < FN; ++jj)
                        acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bq[ks][jj], af[ks][i], acc[i][jj], 0, 0, 0);
        };
        if (nks == 8) mma_all(std::integral_constant<int, 8>{});
        else if (nks == 4) mma_all(std::integral_constant<int, 4>{});
        else if (nks == 2) mma_all(std::integral_constant<int, 2>{});
        else {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if (ks < nks) {
                    bf16x8_t af[FM];
#pragma unroll
                    for (int i = 0; i < FM; ++i) af[i] = fragment<false, BM, BK>(sA + (ks >> 1) * SUB, wm * WM + i * 16, ks & 1, g, c16);
#pragma unroll
                    for (int i = 0; i < FM; ++i)
#pragma unroll
                        for (int jj = 0; jj < FN; ++jj)
                            acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bq[ks][jj], af[i], acc[i][jj], 0, 0, 0);
                }
            }
        }
        const bf16_t* sR = sA + res_off;
        const bf16_t* sX = sA + aux_off;
        if (PROF) {
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int jj = 0; jj < FN; ++jj) asm volatile("" : "+v"(acc[i][jj]));
            t1 = cyc_now(); pc[2] += t1 - t0; t0 = t1;
        }
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const int m = tm * BM + wm * WM + i * 16 + c16;
            float v[8];
#pragma unroll
            for (int jj = 0; jj < FN; ++jj)
#pragma unroll
                for (int e = 0; e < 4; ++e) v[jj * 4 + e] = acc[i][jj][e] * csc[jj][e] + csh[jj][e];
            const unsigned long long didx = (unsigned long long)m * N + ncol;
            if (drop == 1) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = dropout_keep(dseed, didx + e, dth) ? v[e] * dsc : 0.f;
            }
            if (has_res) {
                const uint4 r4 = *reinterpret_cast<const uint4*>(sR + rx_off[i]);
                v[0] += __uint_as_float(r4.x << 16); v[1] += __uint_as_float(r4.x & 0xffff0000u);
                v[2] += __uint_as_float(r4.y << 16); v[3] += __uint_as_float(r4.y & 0xffff0000u);
                v[4] += __uint_as_float(r4.z << 16); v[5] += __uint_as_float(r4.z & 0xffff0000u);
                v[6] += __uint_as_float(r4.w << 16); v[7] += __uint_as_float(r4.w & 0xffff0000u);
            }
            if (has_aux) {
                const uint4 x4 = *reinterpret_cast<const uint4*>(sX + rx_off[i]);
                v[0] = __uint_as_float(x4.x << 16) > 0.f ? v[0] : 0.f; v[1] = __uint_as_float(x4.x & 0xffff0000u) > 0.f ? v[1] : 0.f;
                v[2] = __uint_as_float(x4.y << 16) > 0.f ? v[2] : 0.f; v[3] = __uint_as_float(x4.y & 0xffff0000u) > 0.f ? v[3] : 0.f;
                v[4] = __uint_as_float(x4.z << 16) > 0.f ? v[4] : 0.f; v[5] = __uint_as_float(x4.z & 0xffff0000u) > 0.f ? v[5] : 0.f;
                v[6] = __uint_as_float(x4.w << 16) > 0.f ? v[6] : 0.f; v[7] = __uint_as_float(x4.w & 0xffff0000u) > 0.f ? v[7] : 0.f;
            }
            if (drop == 2) {
                if (ACT == TOIST_ACT_RELU) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = dropout_keep(dseed, didx + e, dth) ? v[e] * dsc : 0.f;
            }
            u32x4_t o = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7])};
            if (ACT == TOIST_ACT_RELU && drop != 2) {
                // ReLU on the packed bf16 pairs (one v_pk_max_i16 per pair instead of a v_max_f32 per value): rounding is monotone and
                // keeps the sign, so max(round(v), 0) = round(max(v, 0)); a negative (or -0) half has its int16 sign bit set
                // (written as asm: hipcc's SLP pass merged two __builtin_elementwise_max calls on <2 x i16> into one and fed its result
                // to BOTH words -- seen in the ISA, wrong values in tools/r3/panel2.py)
                unsigned r0, r1, r2, r3;
                asm("v_pk_max_i16 %0, %1, 0" : "=v"(r0) : "v"(o[0]));
                asm("v_pk_max_i16 %0, %1, 0" : "=v"(r1) : "v"(o[1]));
                asm("v_pk_max_i16 %0, %1, 0" : "=v"(r2) : "v"(o[2]));
                asm("v_pk_max_i16 %0, %1, 0" : "=v"(r3) : "v"(o[3]));
                o[0] = r0; o[1] = r1; o[2] = r2; o[3] = r3;
            }
            store16_asm(rsC, (col_ok && m < M) ? (m * ldc + ncol) * 2 : OOB, o);
        }
        if (PROF) { t1 = cyc_now(); pc[3] += t1 - t0; t0 = t1; }
        if (++slot == ns) slot = 0;
    }
    if (PROF && tid == 0 && p.workspace != nullptr) {
        float* o = p.workspace + (size_t)blockIdx.x * 8;
        o[0] = (float)pc[0]; o[1] = (float)pc[1]; o[2] = (float)pc[2]; o[3] = (float)pc[3]; o[4] = (float)T;
    }
}

// Tile code 135, or picked by the dispatcher (panel_min_tiles) when the call qualifies.
static bool panel_applies(const toist_gemm& d) {
    if (d.a_kind != TOIST_A_ROWK || (d.b_kind != TOIST_B_ROWK && d.b_kind != TOIST_B_KROW)) return false;
    if (d.b_kind == TOIST_B_KROW && d.b.kin > 0) return false;
    if (d.K > 256 || (d.K % 8) != 0 || (d.a.ld % 8) != 0 || (d.b.ld % 8) != 0) return false;
    if (d.batch != 1 || d.split_k != 1 || d.group != nullptr || d.a2 != nullptr || d.a_colsum != nullptr || (d.flags & 1)) return false;
    if ((d.N + 63) / 64 > 64 || (d.M + 63) / 64 < 8) return false;      // row tiles are dealt to 8 XCDs: too few would leave XCDs idle
    if ((long long)d.M * d.a.ld >= (1ll << 30) || (long long)d.K * d.b.ld >= (1ll << 30) || (long long)d.N * d.b.ld >= (1ll << 30)) return false;   // 32-bit element offsets
    // the kernel's lean epilogue: bf16 rows in whole 16-byte chunks, per-column scale / shift, residual, {none, ReLU, aux > 0 mask}
    const toist_epilogue& e = d.epi;
    if (e.out_f32 || e.accumulate || e.rscale || e.pre_out || e.cmap || e.res_div > 0) return false;
    if (e.drop_where && !(e.drop_p >= 0.f && e.drop_p < 1.f)) return false;
    if (e.act != TOIST_ACT_NONE && e.act != TOIST_ACT_RELU && e.act != TOIST_ACT_MASK_POS) return false;
    if ((d.N % 8) != 0 || (d.ldc % 8) != 0 || !aligned16(d.c) || (long long)d.M * d.ldc >= (1ll << 31)) return false;
    if (e.res && ((e.ldr % 8) != 0 || !aligned16(e.res) || (long long)d.M * e.ldr >= (1ll << 31))) return false;
    if (e.act == TOIST_ACT_MASK_POS && ((e.ldaux % 8) != 0 || !aligned16(e.aux) || (long long)d.M * e.ldaux >= (1ll << 31))) return false;
    if ((e.scale && (((size_t)e.scale) & 15)) || (e.shift && (((size_t)e.shift) & 15))) return false;
    return true;
}

static long long panel_min_tiles() {   // (32 -- the 800-query decoder linears, 52 tiles -- measured 559.8 -> 563.8 images/s over two noisy rounds: not taken)
    static const long long v = tuning_knob("TOIST_PANEL_MIN_TILES", 128);
    return v;
}

static int launch_panel(const toist_gemm& d, hipStream_t st) {
    const int nt_m = (d.M + 63) / 64, nt_n = (d.N + 63) / 64;
    const int m_per = (nt_m + 7) / 8;
    int groups = 64 / nt_n;                       // two workgroups per CU, 32 CUs per XCD
    if (groups > m_per) groups = m_per;
    if (groups < 1) groups = 1;
    dim3 grid(8u * (unsigned)(groups * nt_n), 1, 1);
    const bool rowk = d.b_kind == TOIST_B_ROWK;
#define TOIST_PANEL(ACT)                                                                                             \
    do {                                                                                                             \
        if (rowk) hipLaunchKernelGGL((panel_kernel<TOIST_B_ROWK, ACT>), grid, dim3(256), 0, st, d);                  \
        else hipLaunchKernelGGL((panel_kernel<TOIST_B_KROW, ACT>), grid, dim3(256), 0, st, d);                       \
    } while (0)
    switch (d.epi.act) {
        case TOIST_ACT_RELU: TOIST_PANEL(TOIST_ACT_RELU); break;
        case TOIST_ACT_MASK_POS: TOIST_PANEL(TOIST_ACT_MASK_POS); break;
        default: TOIST_PANEL(TOIST_ACT_NONE); break;
    }
#undef TOIST_PANEL
    return TOIST_OK;
}

// panel2_kernel: what panel_applies admits, with byte offsets that fit the 2 GiB buffer descriptors of the DMA / store path
static bool panel2_applies(const toist_gemm& d) {
    if (!panel_applies(d)) return false;
    const long long lim = 1ll << 30;
    if ((long long)d.M * d.ldc >= lim) return false;
    if (d.epi.res && (long long)d.M * d.epi.ldr >= lim) return false;
    if (d.epi.act == TOIST_ACT_MASK_POS && (long long)d.M * d.epi.ldaux >= lim) return false;
    return true;
}

// variant = ring code of the tile word (tile >> 8): low nibble = ring stages (2..4; 0 = pick), bit 4 = 32-row blocks
static int launch_panel2(const toist_gemm& d, int variant, hipStream_t st) {
    int ns = variant & 15;
    int bm = (variant & 16) ? 32 : 64;
    const int kt = (d.K + 63) / 64;
    const bool has_res = d.epi.res != nullptr, has_aux = d.epi.act == TOIST_ACT_MASK_POS;
    if (ns == 0) {
        // measured (tools/r3/panel2.py, MI355X): two ring stages everywhere (a deeper ring never paid: the tile loop is bound by its own
        // instruction stream, not by load latency); 64-row blocks while two workgroups of them fit a CU's LDS (80 KB each: no mask tile at
        // K = 256), 32-row blocks otherwise (three workgroups per CU)
        ns = 2;
        bm = (2 * 64 * 64 * 2 * (kt + (has_res ? 1 : 0) + (has_aux ? 1 : 0)) <= 80 * 1024) ? 64 : 32;
    }
    if (ns < 2 || ns > 4) { set_last_error("toist_gemm_bf16: panel ring of %d stages (2..4)", ns); return TOIST_EINVAL; }
    const int stage = bm * 64 * 2 * (kt + (has_res ? 1 : 0) + (has_aux ? 1 : 0));
    int lds = P2_HEAD + ns * stage;
    if (lds < 32768) lds = 32768;                 // the k-major weight panel is staged through 32 KB once
    if (lds > 160 * 1024) { set_last_error("toist_gemm_bf16: panel ring of %d x %d bytes exceeds the LDS", ns, stage); return TOIST_EINVAL; }
    const int nt_m = (d.M + bm - 1) / bm, nt_n = (d.N + 63) / 64;
    const int m_per = (nt_m + 7) / 8;
    int per_cu = (160 * 1024) / lds;              // workgroups a CU holds (LDS); registers allow 2 waves per SIMD
    const int max_per_cu = bm == 64 ? 2 : 3;      // registers: 2 (164 VGPRs) / 3 (136) waves per SIMD
    if (per_cu > max_per_cu) per_cu = max_per_cu;
    int groups = (32 * per_cu) / nt_n;
    if (groups > m_per) groups = m_per;
    if (groups < 1) groups = 1;
    dim3 grid(8u * (unsigned)(groups * nt_n), 1, 1);
    const bool rowk = d.b_kind == TOIST_B_ROWK;
#define TOIST_PANEL2_ONE(BKD, ACT, BM_)                                                                                                   \
    do {                                                                                                                                  \
        static std::atomic<unsigned long long> done{0};                                                                                   \
        if (!lds_attr_once_flag(done, [] { return hipFuncSetAttribute((const void*)panel2_kernel<BKD, ACT, BM_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess; })) { \
            set_last_error("toist_gemm_bf16: cannot enable 160 KB of LDS for the panel kernel");                                          \
            return TOIST_EHIP;                                                                                                            \
        }                                                                                                                                 \
        hipLaunchKernelGGL((panel2_kernel<BKD, ACT, BM_>), grid, dim3(256), lds, st, d, ns);                                              \
    } while (0)
#define TOIST_PANEL2(ACT)                                                                                             \
    do {                                                                                                              \
        if (rowk) { if (bm == 64) TOIST_PANEL2_ONE(TOIST_B_ROWK, ACT, 64); else TOIST_PANEL2_ONE(TOIST_B_ROWK, ACT, 32); } \
        else { if (bm == 64) TOIST_PANEL2_ONE(TOIST_B_KROW, ACT, 64); else TOIST_PANEL2_ONE(TOIST_B_KROW, ACT, 32); }      \
    } while (0)
    if ((d.flags & 2048) && d.workspace != nullptr && rowk && d.epi.act == TOIST_ACT_RELU) {      // experiments: per-phase cycle counters
        if (hipFuncSetAttribute((const void*)panel2_kernel<TOIST_B_ROWK, TOIST_ACT_RELU, 64, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipFuncSetAttribute((const void*)panel2_kernel<TOIST_B_ROWK, TOIST_ACT_RELU, 32, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return TOIST_EHIP;
        if (bm == 64) hipLaunchKernelGGL((panel2_kernel<TOIST_B_ROWK, TOIST_ACT_RELU, 64, true>), grid, dim3(256), lds, st, d, ns);
        else hipLaunchKernelGGL((panel2_kernel<TOIST_B_ROWK, TOIST_ACT_RELU, 32, true>), grid, dim3(256), lds, st, d, ns);
        return TOIST_OK;
    }
    switch (d.epi.act) {
        case TOIST_ACT_RELU: TOIST_PANEL2(TOIST_ACT_RELU); break;
        case TOIST_ACT_MASK_POS: TOIST_PANEL2(TOIST_ACT_MASK_POS); break;
        default: TOIST_PANEL2(TOIST_ACT_NONE); break;
    }
#undef TOIST_PANEL2
#undef TOIST_PANEL2_ONE
    return TOIST_OK;
}

// true when the halo kernel covers this call (the dispatcher then never looks at the tile code)
static bool conv3_applies(const toist_gemm& d) {
    const bool fwd = d.a_kind == TOIST_A_CONV && d.b_kind == TOIST_B_ROWK;
    const bool dgr = d.a_kind == TOIST_A_CONVT && d.b_kind == TOIST_B_KROW;
    if (!fwd && !dgr) return false;
    const toist_operand& a = d.a;
    if (a.R != 3 || a.S != 3 || a.stride != 1 || a.pad != 1 || a.dil != 1) return false;
    if (a.PH != a.SH || a.PW != a.SW || (a.SC % C3_BK) != 0 || d.K != 9 * a.SC) return false;
    if (C3_BM + 2 * (a.SW + 1) > C3_NP * 8) return false;            // patch rows must fit the 28 pieces
    if (d.batch != 1 || d.split_k != 1 || d.a_colsum != nullptr) return false;
    if ((long long)d.M * a.SC >= (1ll << 30)) return false;           // 32-bit element offsets
    if (fwd && d.b.ld != d.K) return false;
    if (dgr && (d.b.kin != a.SC || (d.N % 8) != 0)) return false;
    return d.M >= 2048;                                                // tiny grids: the generic 64x64 tiles fill more CUs
}

#ifndef GEMM_UNIT
static int clamp_split(int split_k, int K, int tile);
// gemm128_kernel (tile code 136): one problem, K a multiple of 64 with at least 4 k-tiles, bf16 rows finished by {alpha, scale, shift,
// residual, ReLU | aux > 0 mask}.  Operands: row-major A with row-major or plain k-major B (1x1 convolutions, nn.Linear and their data
// gradients); convolution gather of stride 1 or 2 with row-major weights (forward); its stride-1 transposed gather on a same-size plane with the
// two-level k-major weights (data gradient) -- up to 16 taps, source channels a multiple of 64.
static bool gemm128_applies(const toist_gemm& d) {
    const bool plain = d.a_kind == TOIST_A_ROWK && (d.b_kind == TOIST_B_ROWK || (d.b_kind == TOIST_B_KROW && d.b.kin == 0));
    const bool fwd = d.a_kind == TOIST_A_CONV && d.b_kind == TOIST_B_ROWK;
    const bool dgr = d.a_kind == TOIST_A_CONVT && d.b_kind == TOIST_B_KROW;
    if (!plain && !fwd && !dgr) return false;
    if (d.K < 256 || (d.K % 64) != 0 || (d.b.ld % 8) != 0) return false;
    if (d.batch != 1 || d.split_k != 1 || d.group != nullptr || d.a2 != nullptr || d.a_colsum != nullptr || (d.flags & 1)) return false;
    const long long lim = 1ll << 30;
    if ((long long)d.K * d.b.ld >= lim || (long long)d.N * d.b.ld >= lim) return false;       // 32-bit byte offsets
    if (plain) {
        if ((d.a.ld % 8) != 0 || (long long)d.M * d.a.ld >= lim) return false;
    } else {
        const toist_operand& a = d.a;
        if ((a.stride != 1 && !fwd) || a.stride < 1 || a.stride > 2 || a.R * a.S > 16 || (a.SC % 64) != 0 || d.K != a.R * a.S * a.SC) return false;
        if ((long long)(d.M / (a.PH * a.PW) + 1) * a.SH * a.SW * a.SC >= lim) return false;
        if (fwd && d.b.ld != d.K) return false;
        if (dgr && (a.PH != a.SH || a.PW != a.SW || d.b.kin != a.SC)) return false;
        if (dgr && (a.pad - (a.R - 1) * a.dil > 0 || a.pad - (a.S - 1) * a.dil > 0)) return false;   // the shifted base must not lie beyond tap (0, 0)
    }
    const toist_epilogue& e = d.epi;
    if (!lean_epilogue_ok(d) || e.drop_where || e.cmap) return false;
    if (e.act != TOIST_ACT_NONE && e.act != TOIST_ACT_RELU && e.act != TOIST_ACT_MASK_POS) return false;
    if ((d.N % 8) != 0) return false;                     // ldc / ldr / ldaux % 8 and 16-byte bases: lean_epilogue_ok above
    return true;
}

// Where the dispatcher picks it (profiles/r03_gemm128_us.txt): one block per CU, so it needs about a chip of 128 x 128 tiles and a deep
// reduction to amortise its prologue / k-fold epilogue.  Gathers: every stride-1 3x3 of layers 2-4 at the bench batch (-1 .. -16 us per
// launch against the 64 x 64 tiles and the halo kernel).  Plain GEMMs: K >= 768 from 150 tiles on (12800 x 256 x 1024: 14.7 / 15.5 vs
// 18.2 / 20.3 us forward / data gradient, 12800 x 512 x 1024: 25.4 / 27.5 vs 29.2 / 31.8, 4096^3: 150 vs 169); K = 512 loses (two rounds of
// 8 k-tiles do not amortise the fixed costs), 100 tiles are a draw.
static bool gemm128_pays(const toist_gemm& d) {
    static const int on = (int)tuning_knob("TOIST_GEMM128", 1);
    if (!on || !gemm128_applies(d)) return false;
    const long long tiles = (long long)((d.M + G8_BM - 1) / G8_BM) * ((d.N + G8_BN - 1) / G8_BN);
    if (d.a_kind != TOIST_A_ROWK) return tiles >= 96 && (d.N % G8_BN) == 0;
    return d.K >= 768 && tiles >= 150;
}

template <int NS>
static int launch_gemm128_ring(const toist_gemm& d, hipStream_t st) {
    static std::atomic<unsigned long long> done{0};
    constexpr int LDS = NS * G8_STAGE_BYTES;
    if (!lds_attr_once_flag(done, [] {
            return hipFuncSetAttribute((const void*)gemm128_kernel<TOIST_A_ROWK, TOIST_B_ROWK, NS>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) == hipSuccess &&
                   hipFuncSetAttribute((const void*)gemm128_kernel<TOIST_A_ROWK, TOIST_B_KROW, NS>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) == hipSuccess &&
                   hipFuncSetAttribute((const void*)gemm128_kernel<TOIST_A_CONV, TOIST_B_ROWK, NS>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) == hipSuccess &&
                   hipFuncSetAttribute((const void*)gemm128_kernel<TOIST_A_CONVT, TOIST_B_KROW, NS>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) == hipSuccess;
        })) {
        set_last_error("toist_gemm_bf16: cannot enable %d bytes of LDS for the 128x128 kernel", LDS);
        return TOIST_EHIP;
    }
    const int tiles = ((d.M + G8_BM - 1) / G8_BM) * ((d.N + G8_BN - 1) / G8_BN);
    dim3 grid((tiles + 7) & ~7, 1, 1);
    if (d.a_kind == TOIST_A_CONV) hipLaunchKernelGGL((gemm128_kernel<TOIST_A_CONV, TOIST_B_ROWK, NS>), grid, dim3(512), LDS, st, d);
    else if (d.a_kind == TOIST_A_CONVT) hipLaunchKernelGGL((gemm128_kernel<TOIST_A_CONVT, TOIST_B_KROW, NS>), grid, dim3(512), LDS, st, d);
    else if (d.b_kind == TOIST_B_KROW) hipLaunchKernelGGL((gemm128_kernel<TOIST_A_ROWK, TOIST_B_KROW, NS>), grid, dim3(512), LDS, st, d);
    else hipLaunchKernelGGL((gemm128_kernel<TOIST_A_ROWK, TOIST_B_ROWK, NS>), grid, dim3(512), LDS, st, d);
    return TOIST_OK;
}

static int launch_gemm128(const toist_gemm& d, hipStream_t st) {
    static const int ring = (int)tuning_knob("TOIST_G8_RING", 5);
    return (ring == 5 && d.K >= 5 * G8_BK) ? launch_gemm128_ring<5>(d, st) : launch_gemm128_ring<4>(d, st);
}

// gemm128w_kernel (tile code 137): weight gradients -- k-major A (dy [pixels][Co]), B = k-major rows (1x1) or the stride-1 same-size
// gather with whole 128-column tiles inside a tap; f32 output with alpha / rscale / accumulate, or k-slice partials; grouped or single.
static bool gemm128w_applies(const toist_gemm& d) {
    if (d.a_kind != TOIST_A_KROW) return false;
    const bool plain = d.b_kind == TOIST_B_KROW && d.b.kin == 0, gather = d.b_kind == TOIST_B_CONVX;
    if (!plain && !gather) return false;
    if ((d.K % 64) != 0 || (d.M % 8) != 0 || (d.N % 8) != 0 || (d.a.ld % 8) != 0) return false;
    if (d.batch_inner != 1 || (d.batch != 1 && d.group == nullptr) || d.a2 != nullptr || d.a_colsum != nullptr || (d.flags & (1 | TOIST_GEMM_SPLIT_EPILOGUE))) return false;
    const int split = clamp_split(d.split_k, d.K, 65);                                  // the slices the launch will really make
    const int ktiles = d.K / 64, kper = (ktiles + split - 1) / split;
    if (kper < 4 || (long long)(split - 1) * kper + 4 > ktiles) return false;          // every k-slice holds at least 4 k-tiles
    if (split > 1 && d.workspace == nullptr) return false;
    const long long lim = 1ll << 30;
    if ((long long)d.K * d.a.ld >= lim) return false;                                   // 32-bit byte offsets
    if (plain) {
        if ((d.b.ld % 8) != 0 || (long long)d.K * d.b.ld >= lim) return false;
    } else {
        const toist_operand& b = d.b;
        if (b.stride < 1 || b.stride > 2 || (b.SC % 128) != 0 || d.N != b.R * b.S * b.SC) return false;
        if (b.stride == 1 && (b.PH != b.SH || b.PW != b.SW)) return false;                 // stride 1: the same-size plane trick (address = pixel + constant)
        if ((d.K % (b.PH * b.PW)) != 0) return false;
        if ((long long)(d.K / (b.PH * b.PW) + 1) * b.SH * b.SW * b.SC + (long long)b.SW * (b.pad + 1) * b.SC >= lim) return false;
    }
    const toist_epilogue& e = d.epi;
    if (!e.out_f32 || e.scale || e.shift || e.res || e.pre_out || e.act != TOIST_ACT_NONE || e.drop_where || e.cmap || e.res_div > 0) return false;
    if ((d.ldc % 4) != 0 || (((size_t)d.c) & 15) != 0) return false;
    return true;
}

// where it replaces the grouped / split tiles the host asked for (profiles/r03_gemm128w_us.txt)
static bool gemm128w_pays(const toist_gemm& d) {
    static const int on = (int)tuning_knob("TOIST_GEMM128W", 1);
    if (!on || !gemm128w_applies(d)) return false;
    const long long wgs = (long long)((d.M + G8_BM - 1) / G8_BM) * ((d.N + G8_BN - 1) / G8_BN) * d.batch * clamp_split(d.split_k, d.K, 65);
    return wgs >= 128 && (d.M % 64) == 0 && (d.N % 128) == 0;
}

// gemm256w_kernel (tile code 138): what gemm128w_kernel takes, with M a multiple of 256 and k-slices of at least 6 x 32 rows
static bool gemm256w_applies(const toist_gemm& d) {
    if (!gemm128w_applies(d) || (d.M % G9_BM) != 0 || (d.N % G9_BN) != 0) return false;
    const int split = clamp_split(d.split_k, d.K, 65);
    const int k64 = d.K / 64, kper = (k64 + split - 1) / split;
    return 2 * (k64 - (split - 1) * kper) >= G9_NS;      // the last (shortest) slice fills the ring
}

static bool gemm256w_pays(const toist_gemm& d) {
    static const int on = (int)tuning_knob("TOIST_GEMM256W", 1);
    if (!on || !gemm256w_applies(d)) return false;
    const long long wgs = (long long)(d.M / G9_BM) * (d.N / G9_BN) * d.batch * clamp_split(d.split_k, d.K, 65);
    return wgs >= 128;
}

static int launch_gemm256w(const toist_gemm& d, hipStream_t st) {
    static std::atomic<unsigned long long> done{0};
    if (!lds_attr_once_flag(done, [] {
            return hipFuncSetAttribute((const void*)gemm256w_kernel<TOIST_B_KROW>, hipFuncAttributeMaxDynamicSharedMemorySize, G9_LDS) == hipSuccess &&
                   hipFuncSetAttribute((const void*)gemm256w_kernel<TOIST_B_CONVX>, hipFuncAttributeMaxDynamicSharedMemorySize, G9_LDS) == hipSuccess;
        })) {
        set_last_error("toist_gemm_bf16: cannot enable %d bytes of LDS for the 256x128 weight-gradient kernel", G9_LDS);
        return TOIST_EHIP;
    }
    toist_gemm dd = d;
    dd.split_k = clamp_split(d.split_k, d.K, 65);
    const int tiles = (d.M / G9_BM) * (d.N / G9_BN);
    const int nz = dd.batch * dd.split_k;
    dim3 grid((tiles + 7) & ~7, nz, 1);
    if (nz >= 8) grid = dim3(8 * ((nz + 7) / 8) * tiles, 1, 1);
    if (d.b_kind == TOIST_B_CONVX) hipLaunchKernelGGL((gemm256w_kernel<TOIST_B_CONVX>), grid, dim3(512), G9_LDS, st, dd);
    else hipLaunchKernelGGL((gemm256w_kernel<TOIST_B_KROW>), grid, dim3(512), G9_LDS, st, dd);
    return TOIST_OK;
}

// the big generic tiles the host asks for on (grouped) weight gradients -- and the automatic choice -- give way to the 128x128 kernel
static int wgrad_tile_replaced(const toist_gemm& d) {       // 0 = no, else 137 / 138
    const int t = d.tile & 255;
    if (!((t == 0 || t == 129 || t == 130 || t == 134) && d.a_kind == TOIST_A_KROW)) return 0;
    if (gemm256w_pays(d)) return 138;
    return gemm128w_pays(d) ? 137 : 0;
}

static int launch_gemm128w(const toist_gemm& d, hipStream_t st) {
    static std::atomic<unsigned long long> done{0};
    if (!lds_attr_once_flag(done, [] {
            return hipFuncSetAttribute((const void*)gemm128w_kernel<TOIST_B_KROW, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * G8_STAGE_BYTES) == hipSuccess &&
                   hipFuncSetAttribute((const void*)gemm128w_kernel<TOIST_B_CONVX, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * G8_STAGE_BYTES) == hipSuccess &&
                   hipFuncSetAttribute((const void*)gemm128w_kernel<TOIST_B_KROW, 5>, hipFuncAttributeMaxDynamicSharedMemorySize, 5 * G8_STAGE_BYTES) == hipSuccess &&
                   hipFuncSetAttribute((const void*)gemm128w_kernel<TOIST_B_CONVX, 5>, hipFuncAttributeMaxDynamicSharedMemorySize, 5 * G8_STAGE_BYTES) == hipSuccess;
        })) {
        set_last_error("toist_gemm_bf16: cannot enable %d bytes of LDS for the 128x128 weight-gradient kernel", 5 * G8_STAGE_BYTES);
        return TOIST_EHIP;
    }
    const int tiles = ((d.M + G8_BM - 1) / G8_BM) * ((d.N + G8_BN - 1) / G8_BN);
    toist_gemm dd = d;
    dd.split_k = clamp_split(d.split_k, d.K, 65);
    const int nz = dd.batch * dd.split_k;
    dim3 grid((tiles + 7) & ~7, nz, 1);
    if (nz >= 8) grid = dim3(8 * ((nz + 7) / 8) * tiles, 1, 1);     // pairs pinned to XCDs (see the kernel)
    static const int ring = (int)tuning_knob("TOIST_G8W_RING", 5);
    const int kper = (d.K / 64 + dd.split_k - 1) / dd.split_k;
    const bool five = ring == 5 && (long long)(dd.split_k - 1) * kper + 5 <= d.K / 64;      // every slice holds a ring's worth of k-tiles
    if (d.b_kind == TOIST_B_CONVX) {
        if (five) hipLaunchKernelGGL((gemm128w_kernel<TOIST_B_CONVX, 5>), grid, dim3(512), 5 * G8_STAGE_BYTES, st, dd);
        else hipLaunchKernelGGL((gemm128w_kernel<TOIST_B_CONVX, 4>), grid, dim3(512), 4 * G8_STAGE_BYTES, st, dd);
    } else {
        if (five) hipLaunchKernelGGL((gemm128w_kernel<TOIST_B_KROW, 5>), grid, dim3(512), 5 * G8_STAGE_BYTES, st, dd);
        else hipLaunchKernelGGL((gemm128w_kernel<TOIST_B_KROW, 4>), grid, dim3(512), 4 * G8_STAGE_BYTES, st, dd);
    }
    return TOIST_OK;
}

#endif  // GEMM_UNIT (128x128 kernels, host side)

static int launch_conv3(const toist_gemm& d, hipStream_t st) {
    // > 64 KiB of dynamic LDS has to be enabled per kernel and per device (idempotent)
    if (!lds_attr_once(0, [] {
            return hipFuncSetAttribute((const void*)conv3_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, C3_LDS) == hipSuccess &&
                   hipFuncSetAttribute((const void*)conv3_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, C3_LDS) == hipSuccess &&
                   hipFuncSetAttribute((const void*)conv3_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, C3_LDS) == hipSuccess &&
                   hipFuncSetAttribute((const void*)conv3_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, C3_LDS) == hipSuccess;
        })) {
        set_last_error("toist_gemm_bf16: cannot enable %d bytes of LDS for the 3x3 kernel", C3_LDS);
        return TOIST_EHIP;
    }
    const int tiles = ((d.M + C3_BM - 1) / C3_BM) * ((d.N + C3_BN - 1) / C3_BN);
    dim3 grid((tiles + 7) & ~7, 1, 1);
    const bool lean = lean_epilogue_ok(d);
    if (d.a_kind == TOIST_A_CONVT) {
        if (lean) hipLaunchKernelGGL((conv3_kernel<true, true>), grid, dim3(256), C3_LDS, st, d);
        else hipLaunchKernelGGL((conv3_kernel<true, false>), grid, dim3(256), C3_LDS, st, d);
    } else {
        if (lean) hipLaunchKernelGGL((conv3_kernel<false, true>), grid, dim3(256), C3_LDS, st, d);
        else hipLaunchKernelGGL((conv3_kernel<false, false>), grid, dim3(256), C3_LDS, st, d);
    }
    return TOIST_OK;
}

#ifndef GEMM_UNIT   // plain (non-template) kernels live in the main translation unit only
// TOIST_GEMM_SPLIT_EPILOGUE: second half of a split-K GEMM that keeps the complete epilogue.  One thread per 8 output columns of a
// row: the k-slice partials are added in slice order, then the row goes through the same epilogue_row8 as an un-split tile
// (same dropout stream, residual, activation, output type).
__global__ __launch_bounds__(256) void splitk_epilogue_kernel(const toist_gemm p) {
    const int cpr = (p.N + 7) / 8;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)p.M * cpr) return;
    const int m = (int)(idx / cpr), n = (int)(idx - (long long)m * cpr) * 8;
    EpiRow r = epi_row(p, m, n);
    EpiCols cols;
    epilogue_cols(p, n, cols, 0);
    EpiPre pre;
    epilogue_fetch(p, r, 0, pre);
    const size_t MN = (size_t)p.M * p.N;
    const float* q = p.workspace + (size_t)m * p.N + n;
    const bool two = r.nv > 4;                       // N % 4 == 0: a chunk has 4 or 8 valid columns
    float4 lo = make_float4(0.f, 0.f, 0.f, 0.f), hi = lo;
    for (int s = 0; s < p.split_k; ++s) {
        const float4 a = *reinterpret_cast<const float4*>(q + s * MN);
        lo.x += a.x; lo.y += a.y; lo.z += a.z; lo.w += a.w;
        if (two) {
            const float4 b = *reinterpret_cast<const float4*>(q + s * MN + 4);
            hi.x += b.x; hi.y += b.y; hi.z += b.z; hi.w += b.w;
        }
    }
    float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    float xres[8], xaux[8];
    epilogue_operands(p, r, 0, pre, xres, xaux);
    epilogue_row8(p, v, r, 0, 0, xres, xaux, cols);
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

// many split-K reductions in one launch: descriptors travel in the kernel arguments, block -> descriptor by a
// prefix table; a block folds 4096 consecutive elements (16 KiB per slice) of one output
constexpr int RB_MAX = 48, RB_ELEMS = 4096;
struct ReduceBatch {
    toist_reduce_desc d[RB_MAX];
    int blk_end[RB_MAX];
    int n;
};

__global__ __launch_bounds__(256) void splitk_reduce_batch_kernel(const ReduceBatch a) {
    int i = 0;
    while (i + 1 < a.n && (int)blockIdx.x >= a.blk_end[i]) ++i;
    const toist_reduce_desc d = a.d[i];
    const int b0 = (i == 0) ? 0 : a.blk_end[i - 1];
    const long long total = (long long)d.M * d.N, stride = total;
    const long long beg = (long long)(blockIdx.x - b0) * RB_ELEMS;
    for (int q = 0; q < RB_ELEMS / 1024; ++q) {
        const long long e = beg + (long long)(q * 256 + threadIdx.x) * 4;
        if (e >= total) break;
        float4 s4 = *reinterpret_cast<const float4*>(d.ws + e);
        for (int k = 1; k < d.splits; ++k) {
            const float4 t = *reinterpret_cast<const float4*>(d.ws + (size_t)k * stride + e);
            s4.x += t.x; s4.y += t.y; s4.z += t.z; s4.w += t.w;
        }
        const int m = (int)(e / d.N), n = (int)(e - (long long)m * d.N);
        const float f = d.rscale ? d.alpha * d.rscale[m] : d.alpha;
        float* cp = d.out + (long long)m * d.ldc + n;
        if (d.accumulate) { cp[0] += s4.x * f; cp[1] += s4.y * f; cp[2] += s4.z * f; cp[3] += s4.w * f; }
        else { cp[0] = s4.x * f; cp[1] = s4.y * f; cp[2] = s4.z * f; cp[3] = s4.w * f; }
    }
}

// the same for TALL stacks (many slices of a small output: the per-block partial sums of a LayerNorm backward, 416 slices of 256 floats):
// a block folds 64 consecutive elements, 16 threads per float4 column walking the slices 16 apart, then a 16-way sum through LDS
constexpr int RT_ELEMS = 64;
__global__ __launch_bounds__(256) void splitk_reduce_tall_kernel(const ReduceBatch a) {
    __shared__ float4 red[16][16];
    int i = 0;
    while (i + 1 < a.n && (int)blockIdx.x >= a.blk_end[i]) ++i;
    const toist_reduce_desc d = a.d[i];
    const int b0 = (i == 0) ? 0 : a.blk_end[i - 1];
    const long long total = (long long)d.M * d.N;
    const int sl = threadIdx.x >> 4, c = threadIdx.x & 15;
    const long long e = (long long)(blockIdx.x - b0) * RT_ELEMS + c * 4;
    float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (e < total) {
        // four slices in flight per thread (a 208-slice stack was a chain of 13 dependent-latency loads: 13 us for 0.4 MB); the sum keeps its order
        int k = sl;
        for (; k + 48 < d.splits; k += 64) {
            const float4 t0 = *reinterpret_cast<const float4*>(d.ws + (size_t)k * total + e);
            const float4 t1 = *reinterpret_cast<const float4*>(d.ws + (size_t)(k + 16) * total + e);
            const float4 t2 = *reinterpret_cast<const float4*>(d.ws + (size_t)(k + 32) * total + e);
            const float4 t3 = *reinterpret_cast<const float4*>(d.ws + (size_t)(k + 48) * total + e);
            s4.x += t0.x; s4.y += t0.y; s4.z += t0.z; s4.w += t0.w;
            s4.x += t1.x; s4.y += t1.y; s4.z += t1.z; s4.w += t1.w;
            s4.x += t2.x; s4.y += t2.y; s4.z += t2.z; s4.w += t2.w;
            s4.x += t3.x; s4.y += t3.y; s4.z += t3.z; s4.w += t3.w;
        }
        for (; k < d.splits; k += 16) {
            const float4 t = *reinterpret_cast<const float4*>(d.ws + (size_t)k * total + e);
            s4.x += t.x; s4.y += t.y; s4.z += t.z; s4.w += t.w;
        }
    }
    red[sl][c] = s4;
    __syncthreads();
    if (sl == 0 && e < total) {
#pragma unroll
        for (int k = 1; k < 16; ++k) { const float4 t = red[k][c]; s4.x += t.x; s4.y += t.y; s4.z += t.z; s4.w += t.w; }
        const int m = (int)(e / d.N), n = (int)(e - (long long)m * d.N);
        const float f = d.rscale ? d.alpha * d.rscale[m] : d.alpha;
        float* cp = d.out + (long long)m * d.ldc + n;
        if (d.accumulate) { cp[0] += s4.x * f; cp[1] += s4.y * f; cp[2] += s4.z * f; cp[3] += s4.w * f; }
        else { cp[0] = s4.x * f; cp[1] = s4.y * f; cp[2] = s4.z * f; cp[3] = s4.w * f; }
    }
}

#endif  // GEMM_UNIT

// Workgroups of a persistent launch (TOIST_PERSIST_WGS; 0 = one workgroup per tile everywhere).  Applied (launch_variant) only to the
// dispatch-bound launches: row-major A, K <= 256, >= 2048 tiles.  Measured on MI355X (round 2, tools/dbg/gemm_persist.py,
// profiles/r02_gemm_persistent_sweep.txt): 768 workgroups (3 per CU) take 9-16 % off such launches in a back-to-back microbenchmark
// (12800x1024x256: 34.0 -> 29.9 us; 204800x256x64: 88.8 -> 76.0 us); the whole training step gains 1.3 % (435 -> 441 images/s, same
// box).  Applied to EVERY launch the same cap is neutral to negative (1024: -1.4 %).
static long long persist_wgs() {
    static const long long v = [] {
        const long long n = tuning_knob("TOIST_PERSIST_WGS", 768);
        return n <= 0 ? (1LL << 40) : n;
    }();
    return v;
}

// Measured and rejected (round 2, off unless TOIST_GROUP_XCD=1): pinning the problems of a grouped launch to XCDs.  FETCH_SIZE shows the
// grouped layer-3 weight gradients pulling 2-9 x their operand bytes into L2 because every problem's tiles are striped over all eight
// XCDs -- but those re-reads are served by the 256 MB Infinity Cache, not by HBM, and 22 problems on 8 XCDs leave two XCDs idle a
// third of the time: 445.6 vs 449.0 images/s on the same box.
// 64x64x64 launches with at most this many workgroups take the 3-slot ring (TOIST_RING3_MAX_WGS)
static long long ring3_max_wgs() {
    static const long long v = tuning_knob("TOIST_RING3_MAX_WGS", 512);
    return v;
}

static bool xcd_pinned_groups() {
    static const bool v = tuning_knob("TOIST_GROUP_XCD", 0) != 0;
    return v;
}

template <int BM, int BN, int BK, int AK, int BKD>
static int launch_variant(const toist_gemm& d, int ring, hipStream_t st) {
    const int tiles = ((d.M + BM - 1) / BM) * ((d.N + BN - 1) / BN);
    dim3 grid((tiles + 7) & ~7, 1, d.batch * d.split_k);   // 1-D over tiles (XCD-aware order in the kernel), padded to 8
    toist_gemm dd = d;
    if (d.group != nullptr && xcd_pinned_groups()) {       // grouped: problems pinned to XCDs (gemm_kernel), 8 * ceil(Z / 8) * tiles workgroups
        const int Z = d.batch * d.split_k;
        grid = dim3(8u * (unsigned)((Z + 7) / 8) * (unsigned)tiles, 1, 1);
        dd.flags |= 1024;
    }
    constexpr int stage = (BM + BN) * BK * 2;
    // persistent cap (experiments, off by default): the dispatch-bound launches only -- row-major A, short reduction, thousands of tiles
    if (AK == TOIST_A_ROWK && d.K <= 256 && d.batch * d.split_k == 1 && tiles >= 2048) {
        long long cap = persist_wgs() & ~7LL;
        if (cap < 8) cap = 8;
        if ((long long)grid.x > cap) grid.x = (unsigned)cap;
    }
    if (ring == 0) {
        const long long wgs = (long long)tiles * grid.z;
        if (stage <= 8192) ring = 4;
        else if (BM == 64 && BN == 64) ring = wgs <= ring3_max_wgs() ? 3 : 2;
        else ring = 2;
    }
    if (ring * stage > 160 * 1024) {
        set_last_error("toist_gemm_bf16: ring of %d x %d bytes exceeds the LDS", ring, stage);
        return TOIST_EINVAL;
    }
    // instantiated rings (tools/sweep_gemm.py, MI355X): deeper rings never beat 2 slots once the grid fills the chip --
    // the kernel is bound by the LDS-DMA issue rate (~35 B/clk/CU), not by latency; 3 slots help 64x64x64 on small grids
    constexpr bool small = stage <= 8192;
    constexpr bool t65 = (BM == 64 && BN == 64 && BK == 64);
    if (small) ring = 4;
    else if (!t65) ring = 2;
    else if (ring != 2) ring = 3;
    // lean epilogue (epilogue_lean): instantiated for the tiles and operand kinds of the backbone's convolutions
    constexpr bool lean_inst = BK == 64 && ((BM == 64 && BN == 64) || (BM == 64 && BN == 128) || (BM == 128 && BN == 64) || (BM == 128 && BN == 32)) && AK != TOIST_A_KROW;
    if constexpr (lean_inst) {
        if (lean_epilogue_ok(dd)) {
            if constexpr (t65) {
                if (ring == 3) hipLaunchKernelGGL((gemm_kernel<BM, BN, BK, AK, BKD, 3, true>), grid, dim3(256), 0, st, dd);
                else hipLaunchKernelGGL((gemm_kernel<BM, BN, BK, AK, BKD, 2, true>), grid, dim3(256), 0, st, dd);
            } else hipLaunchKernelGGL((gemm_kernel<BM, BN, BK, AK, BKD, 2, true>), grid, dim3(256), 0, st, dd);
            return TOIST_OK;
        }
    }
    if constexpr (small) hipLaunchKernelGGL((gemm_kernel<BM, BN, BK, AK, BKD, 4>), grid, dim3(256), 0, st, dd);
    else if constexpr (t65) {
        if (ring == 3) hipLaunchKernelGGL((gemm_kernel<BM, BN, BK, AK, BKD, 3>), grid, dim3(256), 0, st, dd);
        else hipLaunchKernelGGL((gemm_kernel<BM, BN, BK, AK, BKD, 2>), grid, dim3(256), 0, st, dd);
    } else hipLaunchKernelGGL((gemm_kernel<BM, BN, BK, AK, BKD, 2>), grid, dim3(256), 0, st, dd);
    return TOIST_OK;
}

template <int BM, int BN, int BK>
static int launch_tile(const toist_gemm& d, int ring, hipStream_t st) {
    const int ak = d.a_kind, bk = d.b_kind;
    if (ak == TOIST_A_ROWK && bk == TOIST_B_ROWK) return launch_variant<BM, BN, BK, TOIST_A_ROWK, TOIST_B_ROWK>(d, ring, st);
    else if (ak == TOIST_A_ROWK && bk == TOIST_B_KROW) return launch_variant<BM, BN, BK, TOIST_A_ROWK, TOIST_B_KROW>(d, ring, st);
    else if (ak == TOIST_A_CONV && bk == TOIST_B_ROWK) return launch_variant<BM, BN, BK, TOIST_A_CONV, TOIST_B_ROWK>(d, ring, st);
    else if (ak == TOIST_A_CONVT && bk == TOIST_B_KROW) return launch_variant<BM, BN, BK, TOIST_A_CONVT, TOIST_B_KROW>(d, ring, st);
    else if (ak == TOIST_A_CONV && bk == TOIST_B_KROW) return launch_variant<BM, BN, BK, TOIST_A_CONV, TOIST_B_KROW>(d, ring, st);   // parity classes of a strided dgrad
    else if (ak == TOIST_A_KROW && bk == TOIST_B_KROW) return launch_variant<BM, BN, BK, TOIST_A_KROW, TOIST_B_KROW>(d, ring, st);
    else if (ak == TOIST_A_KROW && bk == TOIST_B_CONVX) return launch_variant<BM, BN, BK, TOIST_A_KROW, TOIST_B_CONVX>(d, ring, st);
    set_last_error("toist_gemm_bf16: unsupported operand kinds a=%d b=%d", ak, bk);
    return TOIST_EINVAL;
}


// ---- translation units ---------------------------------------------------------------------------------------------------
// hipcc needs ~5 minutes for every tile family in one unit.  gemm_tiles_b.hip and gemm_tiles_c.hip include this file with GEMM_UNIT = 1 / 2
// and compile one launch_tiles_*() each (make -j builds the three side by side); everything below this block -- dispatcher, plain kernels,
// the extern "C" entry points -- belongs to the main unit.
int launch_tiles_a(int tile, const toist_gemm& d, int ring, hipStream_t st);   // 64x64x32, 128x128x{32,64}, 128x32x64, 32x128x64
int launch_tiles_b(int tile, const toist_gemm& d, int ring, hipStream_t st);   // 64x64x64
int launch_tiles_c(int tile, const toist_gemm& d, int ring, hipStream_t st);   // 128x64x64, 64x128x64
#if !defined(GEMM_UNIT)
int launch_tiles_a(int tile, const toist_gemm& d, int ring, hipStream_t st) {
    switch (tile) {
        case 64: return launch_tile<64, 64, 32>(d, ring, st);
        case 128: return launch_tile<128, 128, 32>(d, ring, st);
        case 129: return launch_tile<128, 128, 64>(d, ring, st);
        case 132: return launch_tile<128, 32, 64>(d, ring, st);   // narrow N (<= 32 output columns: mask-head convolutions)
        case 133: return launch_tile<32, 128, 64>(d, ring, st);   // narrow M (their weight gradients)
        default: set_last_error("toist_gemm_bf16: bad tile code %d", tile); return TOIST_EINVAL;
    }
}
#elif GEMM_UNIT == 1
int launch_tiles_b(int tile, const toist_gemm& d, int ring, hipStream_t st) { (void)tile; return launch_tile<64, 64, 64>(d, ring, st); }
#elif GEMM_UNIT == 2
int launch_tiles_c(int tile, const toist_gemm& d, int ring, hipStream_t st) {
    if (tile == 130) return launch_tile<128, 64, 64>(d, ring, st);
    return launch_tile<64, 128, 64>(d, ring, st);                  // 134: wide N
}
#endif

#ifndef GEMM_UNIT

// tile code the dispatcher picks for tile == 0
static int auto_tile(const toist_gemm& d) {
    // measured on MI355X (tools/sweep_gemm.py): 128x128x64 only pays once >= ~4 tiles per CU exist and K
    // is deep; below that 64x64x64 tiles keep more workgroups (and DMA) in flight.
    const long long t128 = (long long)((d.M + 127) / 128) * ((d.N + 127) / 128) * (d.batch > 0 ? d.batch : 1) * (d.split_k > 0 ? d.split_k : 1);
    // (round 2: the 128x64 tile beats 128x128 on every large problem measured -- 4096^3: 675 vs 555 TFLOP/s, 8192 x 8192 x 2048: 708 vs 443;
    // 128x128 needs 231-247 VGPRs, two waves per SIMD)
    static const int big_tile = (int)tuning_knob("TOIST_BIG_TILE", 130);
    if (t128 >= 1024 && d.K >= 1024) return big_tile;
    if (d.K > 64 && d.N <= 32 && d.M >= 4096) return 132;   // a 64-wide tile would idle half (or more) of its MFMA columns
    if (d.K > 64 && d.M <= 32 && d.N >= 128) return 133;
    // Wave quantisation of the 64x64 grid: four two-slot workgroups fit a CU (1024 slots; three before the occupancy hint of gemm_kernel,
    // when 800 tiles -- ResNet layer3 at batch 8 -- were a full round plus a 4 % one and this rule bought 28 -> 21 us).  When the last
    // round would be < 1/4 full and there are at most two full ones, 64x128 tiles (half the workgroups, half the A traffic through LDS)
    // finish in fewer rounds: 17000 x 256 x 1024 25.8 -> 21.2 us, 20000 x 256 x 1024 26.6 -> 22.1, 12800 x 384 x 1024 25.9 -> 22.1; at
    // 1.5-1.6 rounds (12800 x 512 x 1024) the 64x64 grid stays ahead (29.3 vs 31.0), as it does below one round (tools/dbg/gemm_tiles.py).
    const long long t64 = (long long)((d.M + 63) / 64) * ((d.N + 63) / 64) * (d.batch > 0 ? d.batch : 1) * (d.split_k > 0 ? d.split_k : 1);
    const long long rounds = t64 / 1024, last = t64 - rounds * 1024;
    static const bool wide = tuning_knob("TOIST_TILE_64x128", 1) != 0;
    if (wide && d.K >= 512 && (d.N % 128) == 0 && rounds >= 1 && rounds <= 2 && last * 4 < 1024 &&
        (d.a_kind == TOIST_A_ROWK || d.a_kind == TOIST_A_CONV) && !d.group)
        return 134;
    // (a nearly full single round -- 800 tiles of 1024 -- stays on 64x64: 64x128 there measured 552 vs 556 images/s for the conv gathers, 546 with row-major A too)
    return (d.K > 64) ? 65 : 64;
}

// drop k-slices that would own no k-tile
static int clamp_split(int split_k, int K, int tile) {
    const int bk0 = (tile == 64 || tile == 128) ? 32 : 64;
    const int ktiles = (K + bk0 - 1) / bk0;
    if (split_k < 1) split_k = 1;
    if (split_k > ktiles) split_k = ktiles;
    const int kper = (ktiles + split_k - 1) / split_k;
    return (ktiles + kper - 1) / kper;
}

#endif  // GEMM_UNIT (dispatcher helpers)

}  // namespace toist

#ifndef GEMM_UNIT
namespace toist {
constexpr int GROUP_MAX = 64;   // 64 x 48 B = 3 KB of kernel arguments
struct GroupArgs { toist_group g[GROUP_MAX]; };
__global__ void group_fill_kernel(const GroupArgs a, int n, toist_group* __restrict__ table) {
    if ((int)threadIdx.x < n) table[threadIdx.x] = a.g[threadIdx.x];
}
}  // namespace toist

// The table travels as kernel arguments (3 KB): no pinned staging buffer to keep alive, and a captured hipGraph replays the
// same pointers from its kernel node.
extern "C" int toist_group_fill(const toist_group* rows, int n, toist_group* table, void* stream) {
    using namespace toist;
    TOIST_REQUIRE(rows != nullptr && table != nullptr && n > 0 && n <= GROUP_MAX, "toist_group_fill: 1..%d entries (got %d)", GROUP_MAX, n);
    GroupArgs a;
    for (int i = 0; i < n; ++i) a.g[i] = rows[i];
    for (int i = n; i < GROUP_MAX; ++i) a.g[i] = toist_group{nullptr, nullptr, 0, 0, 0, 0};
    hipLaunchKernelGGL(group_fill_kernel, dim3(1), dim3(GROUP_MAX), 0, (hipStream_t)stream, a, n, table);
    return check_launch("toist_group_fill");
}

extern "C" int toist_gemm_pick_tile(const toist_gemm* desc) {
    using namespace toist;
    if (desc == nullptr) return 0;
    toist_gemm d = *desc;
    if (d.batch <= 0) d.batch = 1;
    if (d.split_k <= 0) d.split_k = 1;
    if (const int wt = wgrad_tile_replaced(d)) return wt;
    if (desc->tile & 255) return desc->tile & 255;
    if (gemm128_pays(d)) return 136;                    // the order of toist_gemm_bf16's dispatch
    if (d.a_kind == TOIST_A_CONVT && conv3_applies(d)) return 131;
    if ((long long)((d.M + 63) / 64) * ((d.N + 63) / 64) >= panel_min_tiles() && panel_applies(d)) return 135;
    return auto_tile(d);
}

extern "C" int toist_gemm_effective_split(const toist_gemm* desc) {
    using namespace toist;
    if (desc == nullptr) return 0;
    toist_gemm d = *desc;
    if (d.batch <= 0) d.batch = 1;
    if (d.workspace == nullptr) d.workspace = (float*)16;      // asked BEFORE the caller sizes the scratch: "a workspace will be there"
    const int tile = wgrad_tile_replaced(d) ? 137 : (desc->tile & 255) ? (desc->tile & 255) : auto_tile(*desc);     // 137 and 138 slice alike
    return clamp_split(desc->split_k, desc->K, tile);
}

extern "C" int toist_splitk_reduce_batch(const toist_reduce_desc* descs, int n, void* stream) {
    using namespace toist;
    TOIST_REQUIRE(descs != nullptr && n > 0, "toist_splitk_reduce_batch: no descriptors");
    // two passes over the descriptors: ordinary stacks (a thread walks all slices of its 4 elements) and tall ones (> 32 slices of <= 4096 elements)
    for (int tall = 0; tall < 2; ++tall) {
        ReduceBatch a;
        a.n = 0;
        int blocks = 0;
        auto launch = [&]() -> int {
            if (a.n == 0) return TOIST_OK;
            if (tall) hipLaunchKernelGGL(splitk_reduce_tall_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
            else hipLaunchKernelGGL(splitk_reduce_batch_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
            a.n = 0;
            blocks = 0;
            return check_launch("toist_splitk_reduce_batch");
        };
        for (int i = 0; i < n; ++i) {
            const toist_reduce_desc& d = descs[i];
            TOIST_REQUIRE(d.ws && d.out && d.splits >= 1 && d.M > 0 && d.N > 0 && (d.N % 4) == 0 && (d.ldc % 4) == 0 &&
                              ((((size_t)d.ws) | ((size_t)d.out)) & 15) == 0,
                          "toist_splitk_reduce_batch: descriptor %d is malformed", i);
            const long long total = (long long)d.M * d.N;
            const bool is_tall = d.splits >= 8;    // (>= 24: 559 images/s, >= 8: 561, never: 556)   // many slices: 16 threads per float4 column walk them side by side
            if ((int)is_tall != tall) continue;
            a.d[a.n] = d;
            blocks += (int)((total + (tall ? RT_ELEMS : RB_ELEMS) - 1) / (tall ? RT_ELEMS : RB_ELEMS));
            a.blk_end[a.n] = blocks;
            if (++a.n == RB_MAX) {
                const int rc = launch();
                if (rc != TOIST_OK) return rc;
            }
        }
        const int rc = launch();
        if (rc != TOIST_OK) return rc;
    }
    return TOIST_OK;
}

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
    const bool split_epi = (d.flags & TOIST_GEMM_SPLIT_EPILOGUE) != 0 && d.split_k > 1;
    if ((d.split_k > 1 && !split_epi) || d.epi.accumulate)
        TOIST_REQUIRE(d.epi.out_f32, "toist_gemm_bf16: split_k/accumulate needs an f32 output");
    if (split_epi)
        TOIST_REQUIRE(d.workspace != nullptr && d.batch == 1 && d.group == nullptr && (d.N % 4) == 0 && !d.epi.cmap && !d.epi.accumulate &&
                          !(d.flags & TOIST_GEMM_DEFER_REDUCE) && (d.tile & 255) != 131 && !d.a_colsum,
                      "toist_gemm_bf16: TOIST_GEMM_SPLIT_EPILOGUE needs a workspace, batch = 1, no group / row map / accumulate, N %% 4 == 0");
    if (d.a_colsum) TOIST_REQUIRE(d.a_kind == TOIST_A_KROW, "toist_gemm_bf16: a_colsum needs a k-major A operand");
    if (d.group) TOIST_REQUIRE((d.split_k <= 1 || (d.flags & TOIST_GEMM_DEFER_REDUCE)) && d.batch_inner == 1 && !d.epi.cmap && (d.tile & 255) != 131 &&
                                   d.a_kind != TOIST_A_CONVT,
                               "toist_gemm_bf16: a grouped launch takes batch_inner = 1, no cmap, a generic tile, and folds its k-slices through "
                               "toist_splitk_reduce_batch (TOIST_GEMM_DEFER_REDUCE; workspace = [problem][k-slice][M][N])");
    if (d.split_k > 1 && !split_epi) {
        TOIST_REQUIRE(!d.epi.scale && !d.epi.shift && !d.epi.res && d.epi.act == TOIST_ACT_NONE && !d.epi.pre_out && d.epi.drop_where == 0 &&
                          !d.epi.cmap && (d.batch == 1 || d.group != nullptr),
                      "toist_gemm_bf16: split_k only supports alpha/rscale/accumulate epilogues on a single batch (or a grouped launch)");
        TOIST_REQUIRE(d.workspace != nullptr && (d.N % 4) == 0 && (d.ldc % 4) == 0, "toist_gemm_bf16: split_k needs a workspace and N, ldc %% 4 == 0");
    }
    if (d.epi.act >= TOIST_ACT_MASK_POS) TOIST_REQUIRE(d.epi.aux != nullptr, "toist_gemm_bf16: activation %d needs aux", d.epi.act);
    if (d.a2) TOIST_REQUIRE(d.a_kind == TOIST_A_ROWK && aligned16(d.a2) && d.a2_from > 0 && (d.a2_from % 128) == 0 && d.split_k <= 1,
                            "toist_gemm_bf16: a2 needs a row-major A, 16-byte alignment, a2_from %% 128 == 0 and no split");
    if (d.epi.drop_where) TOIST_REQUIRE(d.epi.drop_p >= 0.f && d.epi.drop_p < 1.f, "toist_gemm_bf16: bad dropout p");

    hipStream_t st = (hipStream_t)stream;
    // the 3x3 shared-halo kernel: picked for dgrad (measured 395 vs 351 TFLOP/s on layer 3), explicit tile code 131 otherwise
    // (forward: 384 vs 451 for the generic tiles)
    if (d.tile == 136 || (d.tile == 0 && gemm128_pays(d))) {
        TOIST_REQUIRE(gemm128_applies(d), "toist_gemm_bf16: the 128x128 kernel does not cover this call");
        const int rc8 = launch_gemm128(d, st);
        return rc8 != TOIST_OK ? rc8 : check_launch("toist_gemm_bf16(gemm128)");
    }
    if ((d.tile == 131 || (d.tile == 0 && d.a_kind == TOIST_A_CONVT)) && conv3_applies(d)) {
        const int rc3 = launch_conv3(d, st);
        return rc3 != TOIST_OK ? rc3 : check_launch("toist_gemm_bf16(conv3)");
    }
    TOIST_REQUIRE(d.tile != 131, "toist_gemm_bf16: the 3x3 halo kernel does not cover this call");
    if ((d.tile & 255) == 135 || (d.tile == 0 && (long long)((d.M + 63) / 64) * ((d.N + 63) / 64) >= panel_min_tiles())) {
        if (panel_applies(d)) {
            // tile word 135 | variant << 8: variant 1 = the round-2 kernel, otherwise panel2_kernel (see launch_panel2)
            static const int def_variant = (int)tuning_knob("TOIST_PANEL_VARIANT", 0);
            const int variant = (d.tile >> 8) ? (d.tile >> 8) : def_variant;
            const int rcp = (variant == 1 || !panel2_applies(d)) ? launch_panel(d, st) : launch_panel2(d, variant == 255 ? 0 : variant, st);
            return rcp != TOIST_OK ? rcp : check_launch("toist_gemm_bf16(panel)");
        }
        TOIST_REQUIRE((d.tile & 255) != 135, "toist_gemm_bf16: the short-K panel kernel does not cover this call");
    }
    int tile = d.tile & 255;
    const int ring = d.tile >> 8;   // 0 = pick; else slots of the DMA ring (2..4)
    if (tile == 137) TOIST_REQUIRE(gemm128w_applies(d), "toist_gemm_bf16: the 128x128 weight-gradient kernel does not cover this call");
    if (tile == 138) TOIST_REQUIRE(gemm256w_applies(d), "toist_gemm_bf16: the 256x128 weight-gradient kernel does not cover this call");
    if (const int wt = wgrad_tile_replaced(d)) tile = wt;
    if (tile == 0) tile = auto_tile(d);
    d.split_k = clamp_split(d.split_k, d.K, tile);
    // tile codes: 64 = 64x64x32, 65 = 64x64x64, 128 = 128x128x32, 129 = 128x128x64, 130 = 128x64x64, 132 = 128x32x64, 133 = 32x128x64
    const int bkt = (tile == 64 || tile == 128) ? 32 : 64;   // (134 = 64x128x64)
    if (d.b_kind == TOIST_B_KROW && d.b.kin > 0) TOIST_REQUIRE((d.b.kin % 8) == 0, "toist_gemm_bf16: kin %% 8 != 0");
    (void)bkt;
    int rc;
    switch (tile) {       // tile families by translation unit
        case 65: rc = launch_tiles_b(tile, d, ring, st); break;
        case 130: case 134: rc = launch_tiles_c(tile, d, ring, st); break;
        case 137: rc = launch_gemm128w(d, st); break;
        case 138: rc = launch_gemm256w(d, st); break;
        default: rc = launch_tiles_a(tile, d, ring, st); break;
    }
    if (rc != TOIST_OK) return rc;
    rc = check_launch("toist_gemm_bf16");
    if (rc != TOIST_OK || d.split_k <= 1 || (d.flags & TOIST_GEMM_DEFER_REDUCE)) return rc;
    if (split_epi) {
        const long long chunks = (long long)d.M * ((d.N + 7) / 8);
        hipLaunchKernelGGL(splitk_epilogue_kernel, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, st, d);
        return check_launch("toist_gemm_bf16(splitk epilogue)");
    }
    const long long total4 = ((long long)d.M * d.N) / 4;
    int grid = (int)((total4 + 255) / 256 > 2048 ? 2048 : (total4 + 255) / 256);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(grid), dim3(256), 0, st, (const float*)d.workspace, d.split_k, d.M, d.N, d.epi.alpha,
                       d.epi.rscale, d.epi.accumulate, (float*)d.c, d.ldc);
    return check_launch("toist_gemm_bf16(splitk reduce)");
}
#endif  // GEMM_UNIT (main unit)
