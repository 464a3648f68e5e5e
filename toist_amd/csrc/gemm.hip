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
#include "common.h"

namespace toist {

constexpr int BK = 32;        // k extent of one staged tile (one MFMA k-step)
constexpr int BKP = BK + 8;   // row pitch (elements) of a k-contiguous LDS tile: 80 B, keeps b128 reads spread

__device__ __forceinline__ float gelu_f(float v) { return 0.5f * v * (1.f + erff(v * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad_f(float a) {
    return 0.5f * (1.f + erff(a * 0.70710678118654752f)) + a * 0.3989422804014327f * __expf(-0.5f * a * a);
}

struct PixRow {      // per-thread gather state of one staged row (conv kinds)
    long long base;  // element offset of image n in the source tensor
    int y0, x0;      // CONV: py*stride-pad ; CONVT: py+pad
    bool ok;
};

template <int KIND>
__device__ __forceinline__ PixRow make_pixrow(const toist_operand& o, long long pix, long long npix) {
    PixRow r;
    r.ok = pix < npix;
    const int plane = o.PH * o.PW;
    const int n = (int)(pix / plane);
    const int rem = (int)(pix - (long long)n * plane);
    const int py = rem / o.PW, px = rem - py * o.PW;
    r.base = (long long)n * o.SH * o.SW * o.SC;
    if (KIND == TOIST_A_CONVT) { r.y0 = py + o.pad; r.x0 = px + o.pad; }
    else { r.y0 = py * o.stride - o.pad; r.x0 = px * o.stride - o.pad; }
    return r;
}

// source element offset for (pixel row, tap (r,s)); returns false when the tap falls outside
template <int KIND>
__device__ __forceinline__ bool pix_src(const toist_operand& o, const PixRow& pr, int r, int s, long long& off) {
    int iy, ix;
    if (KIND == TOIST_A_CONVT) {
        const int ty = pr.y0 - r * o.dil, tx = pr.x0 - s * o.dil;
        if (ty < 0 || tx < 0) return false;
        if (o.stride > 1) {
            if ((ty % o.stride) | (tx % o.stride)) return false;
            iy = ty / o.stride; ix = tx / o.stride;
        } else { iy = ty; ix = tx; }
    } else {
        iy = pr.y0 + r * o.dil; ix = pr.x0 + s * o.dil;
        if (iy < 0 || ix < 0) return false;
    }
    if (iy >= o.SH || ix >= o.SW) return false;
    off = pr.base + ((long long)iy * o.SW + ix) * o.SC;
    return true;
}

template <int BM, int BN, int AK, int BKD, bool TR>
__global__ __launch_bounds__(256) void gemm_kernel(const toist_gemm p) {
    constexpr int WM = BM / 2, WN = BN / 2, FM = WM / 16, FN = WN / 16;
    constexpr int ACH = BM * BK / 8 / 256, BCH = BN * BK / 8 / 256;  // 16-byte chunks per thread
    constexpr bool A_KM = (AK == TOIST_A_KROW);    // A staged k-major
    constexpr bool B_KM = (BKD != TOIST_B_ROWK);   // B staged k-major
    constexpr int LDA_T = BM + 8, LDB_T = BN + 8;  // k-major LDS pitches
    constexpr int SA_ELEMS = (A_KM && TR) ? BK * LDA_T : BM * BKP;
    constexpr int SB_ELEMS = (B_KM && TR) ? BK * LDB_T : BN * BKP;
    __shared__ __attribute__((aligned(16))) bf16_t smem[SA_ELEMS + SB_ELEMS];
    bf16_t* sA = smem;
    bf16_t* sB = smem + SA_ELEMS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int g = lane >> 4, c16 = lane & 15;

    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int z = blockIdx.z;
    const int bz = z / p.split_k, ksl = z - bz * p.split_k;
    const int bo = bz / p.batch_inner, bi = bz - bo * p.batch_inner;
    const bf16_t* Ab = (const bf16_t*)p.a.ptr + bo * p.a.bs_outer + bi * p.a.bs_inner;
    const bf16_t* Bb = (const bf16_t*)p.b.ptr + bo * p.b.bs_outer + bi * p.b.bs_inner;
    const long long coff = bo * p.cs_outer + bi * p.cs_inner;

    const int M = p.M, N = p.N, K = p.K;
    const int ktiles = (K + BK - 1) / BK;
    const int kper = (ktiles + p.split_k - 1) / p.split_k;
    const int kt_beg = ksl * kper;
    const int kt_end = (kt_beg + kper < ktiles) ? kt_beg + kper : ktiles;
    if (kt_beg >= kt_end) return;

    // ---- per-thread staging coordinates --------------------------------------------------
    int a_row[ACH], a_kc[ACH];
    PixRow a_pix[ACH];
#pragma unroll
    for (int it = 0; it < ACH; ++it) {
        const int ch = tid + 256 * it;
        if (A_KM) { a_row[it] = ch / (BM / 8); a_kc[it] = ch % (BM / 8); }   // (k row, m chunk)
        else { a_row[it] = ch >> 2; a_kc[it] = ch & 3; }                       // (m row, k chunk)
        if (AK == TOIST_A_CONV || AK == TOIST_A_CONVT) a_pix[it] = make_pixrow<AK>(p.a, (long long)m0 + a_row[it], M);
    }
    int b_row[BCH], b_kc[BCH], b_r[BCH], b_s[BCH], b_c[BCH];
#pragma unroll
    for (int it = 0; it < BCH; ++it) {
        const int ch = tid + 256 * it;
        if (B_KM) { b_row[it] = ch / (BN / 8); b_kc[it] = ch % (BN / 8); }
        else { b_row[it] = ch >> 2; b_kc[it] = ch & 3; }
        if (BKD == TOIST_B_CONVX) {
            const int nn = n0 + b_kc[it] * 8;
            const int tap = nn / p.b.SC;
            b_c[it] = nn - tap * p.b.SC;
            b_r[it] = tap / p.b.S;
            b_s[it] = tap - b_r[it] * p.b.S;
        }
    }

    uint4 ra[ACH], rb[BCH];
    const uint4 zero4 = make_uint4(0, 0, 0, 0);

    auto load_tiles = [&](int kt) {
        const int k0 = kt * BK;
        // ---- A ----
#pragma unroll
        for (int it = 0; it < ACH; ++it) {
            const bf16_t* src = nullptr;
            if (AK == TOIST_A_ROWK) {
                const int m = m0 + a_row[it], kk = k0 + a_kc[it] * 8;
                if (m < M && kk < K) src = Ab + (long long)m * p.a.ld + kk;
            } else if (AK == TOIST_A_KROW) {
                const int k = k0 + a_row[it], m = m0 + a_kc[it] * 8;
                if (k < K && m < M) src = Ab + (long long)k * p.a.ld + m;
            } else {
                int tap, c0;
                if (p.a.SC % BK == 0) { tap = k0 / p.a.SC; c0 = k0 - tap * p.a.SC + a_kc[it] * 8; }
                else { const int kk = k0 + a_kc[it] * 8; tap = kk / p.a.SC; c0 = kk - tap * p.a.SC; }
                if (a_pix[it].ok && tap < p.a.R * p.a.S) {
                    const int r = tap / p.a.S, s = tap - r * p.a.S;
                    long long off;
                    if (pix_src<AK>(p.a, a_pix[it], r, s, off)) src = Ab + off + c0;
                }
            }
            ra[it] = src ? *reinterpret_cast<const uint4*>(src) : zero4;
        }
        // ---- B ----
#pragma unroll
        for (int it = 0; it < BCH; ++it) {
            const bf16_t* src = nullptr;
            if (BKD == TOIST_B_ROWK) {
                const int n = n0 + b_row[it], kk = k0 + b_kc[it] * 8;
                if (n < N && kk < K) src = Bb + (long long)n * p.b.ld + kk;
            } else if (BKD == TOIST_B_KROW) {
                const int k = k0 + b_row[it], n = n0 + b_kc[it] * 8;
                if (k < K && n < N) {
                    if (p.b.kin > 0) {
                        const int tap = k0 / p.b.kin;
                        src = Bb + (long long)(k - tap * p.b.kin) * p.b.ld + (long long)tap * p.b.tap_stride + n;
                    } else src = Bb + (long long)k * p.b.ld + n;
                }
            } else {  // CONVX: k = output pixel, n = (tap, c)
                const long long pix = (long long)k0 + b_row[it];
                const int nn = n0 + b_kc[it] * 8;
                if (nn < N) {
                    const PixRow pr = make_pixrow<TOIST_A_CONV>(p.b, pix, K);
                    long long off;
                    if (pr.ok && pix_src<TOIST_A_CONV>(p.b, pr, b_r[it], b_s[it], off)) src = Bb + off + b_c[it];
                }
            }
            rb[it] = src ? *reinterpret_cast<const uint4*>(src) : zero4;
        }
    };

    auto store_tiles = [&]() {
#pragma unroll
        for (int it = 0; it < ACH; ++it) {
            if (!A_KM) *reinterpret_cast<uint4*>(&sA[a_row[it] * BKP + a_kc[it] * 8]) = ra[it];
            else if (TR) *reinterpret_cast<uint4*>(&sA[a_row[it] * LDA_T + a_kc[it] * 8]) = ra[it];
            else {
                const bf16_t* e = reinterpret_cast<const bf16_t*>(&ra[it]);
#pragma unroll
                for (int j = 0; j < 8; ++j) sA[(a_kc[it] * 8 + j) * BKP + a_row[it]] = e[j];
            }
        }
#pragma unroll
        for (int it = 0; it < BCH; ++it) {
            if (!B_KM) *reinterpret_cast<uint4*>(&sB[b_row[it] * BKP + b_kc[it] * 8]) = rb[it];
            else if (TR) *reinterpret_cast<uint4*>(&sB[b_row[it] * LDB_T + b_kc[it] * 8]) = rb[it];
            else {
                const bf16_t* e = reinterpret_cast<const bf16_t*>(&rb[it]);
#pragma unroll
                for (int j = 0; j < 8; ++j) sB[(b_kc[it] * 8 + j) * BKP + b_row[it]] = e[j];
            }
        }
    };

    // fragment = 8 consecutive k (k = 8*g + j) of tile row (r0 + c16)
    auto frag_rowk = [&](const bf16_t* s, int r0) -> bf16x8_t {
        return *reinterpret_cast<const bf16x8_t*>(&s[(r0 + c16) * BKP + g * 8]);
    };
    auto frag_tr = [&](const bf16_t* s, int ld, int r0) -> bf16x8_t {
        // 16-lane group g transposes the [4 k][16 rows] blocks at k = 8g and k = 8g+4
        const bf16_t* q = &s[(8 * g + (c16 >> 2)) * ld + r0 + (c16 & 3) * 4];
        typedef __attribute__((address_space(3))) s16x4_t* lds_v4;
        const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(q));
        const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(q + 4 * ld));
        union { struct { s16x4_t a, b; } h; bf16x8_t v; } u;
        u.h.a = lo; u.h.b = hi;
        return u.v;
    };

    f32x4_t acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    load_tiles(kt_beg);
    for (int kt = kt_beg; kt < kt_end; ++kt) {
        store_tiles();
        __syncthreads();
        if (kt + 1 < kt_end) load_tiles(kt + 1);
        bf16x8_t af[FM], bfr[FN];
#pragma unroll
        for (int i = 0; i < FM; ++i)
            af[i] = (A_KM && TR) ? frag_tr(sA, LDA_T, wm * WM + i * 16) : frag_rowk(sA, wm * WM + i * 16);
#pragma unroll
        for (int j = 0; j < FN; ++j)
            bfr[j] = (B_KM && TR) ? frag_tr(sB, LDB_T, wn * WN + j * 16) : frag_rowk(sB, wn * WN + j * 16);
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
        __syncthreads();
    }

    // ---- epilogue: lane owns output row m (c16) and 4 consecutive columns n (4*g .. 4*g+3) ----
    const toist_epilogue& e = p.epi;
    const bool atomic = e.accumulate || p.split_k > 1;
    const unsigned drop_thresh = (e.drop_where != 0) ? (unsigned)(e.drop_p * 4294967296.0) : 0u;
    const float drop_scale = (e.drop_where != 0) ? 1.f / (1.f - e.drop_p) : 1.f;
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const int m = m0 + wm * WM + i * 16 + c16;
        if (m >= M) continue;
        long long crow = m;
        if (e.cmap) {
            const int plane = e.cOH * e.cOW;
            const int n_img = m / plane, rem = m - n_img * plane;
            const int oy = rem / e.cOW, ox = rem - oy * e.cOW;
            crow = ((long long)n_img * e.cH + (long long)oy * e.cst) * e.cW + (long long)ox * e.cst;
        }
        const float rs = e.rscale ? e.alpha * e.rscale[m] : e.alpha;
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            const int n = n0 + wn * WN + j * 16 + g * 4;
            if (n >= N) continue;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r] * rs;
            const int nv = (N - n < 4) ? (N - n) : 4;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (r >= nv) break;
                if (e.scale) v[r] *= e.scale[n + r];
                if (e.shift) v[r] += e.shift[n + r];
            }
            if (e.drop_where == 1) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const unsigned long long idx = ((unsigned long long)bz * M + m) * N + n + r;
                    v[r] = dropout_keep(e.drop_seed, idx, drop_thresh) ? v[r] * drop_scale : 0.f;
                }
            }
            if (e.res) {
                const bf16_t* rp = (const bf16_t*)e.res + coff + crow * e.ldr + n;
#pragma unroll
                for (int r = 0; r < 4; ++r) if (r < nv) v[r] += bf2f(rp[r]);
            }
            if (e.pre_out) {
                bf16_t* pp = (bf16_t*)e.pre_out + coff + crow * p.ldc + n;
#pragma unroll
                for (int r = 0; r < 4; ++r) if (r < nv) pp[r] = f2bf(v[r]);
            }
            if (e.act != TOIST_ACT_NONE) {
                float ax[4] = {0.f, 0.f, 0.f, 0.f};
                if (e.act >= TOIST_ACT_MASK_POS) {
                    const bf16_t* ap = (const bf16_t*)e.aux + coff + crow * e.ldaux + n;
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (r < nv) ax[r] = bf2f(ap[r]);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    switch (e.act) {
                        case TOIST_ACT_RELU: v[r] = fmaxf(v[r], 0.f); break;
                        case TOIST_ACT_GELU: v[r] = gelu_f(v[r]); break;
                        case TOIST_ACT_SIGMOID: v[r] = 1.f / (1.f + __expf(-v[r])); break;
                        case TOIST_ACT_MASK_POS: v[r] = ax[r] > 0.f ? v[r] : 0.f; break;
                        case TOIST_ACT_GELU_BWD: v[r] *= gelu_grad_f(ax[r]); break;
                        case TOIST_ACT_SIGMOID_BWD: v[r] *= ax[r] * (1.f - ax[r]); break;
                        default: break;
                    }
                }
            }
            if (e.drop_where == 2) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const unsigned long long idx = ((unsigned long long)bz * M + m) * N + n + r;
                    v[r] = dropout_keep(e.drop_seed, idx, drop_thresh) ? v[r] * drop_scale : 0.f;
                }
            }
            if (e.out_f32) {
                float* cp = (float*)p.c + coff + crow * p.ldc + n;
                if (atomic) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (r < nv) atomicAdd(cp + r, v[r]);
                } else if (nv == 4 && ((((size_t)cp) & 15) == 0)) {
                    *reinterpret_cast<float4*>(cp) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (r < nv) cp[r] = v[r];
                }
            } else {
                bf16_t* cp = (bf16_t*)p.c + coff + crow * p.ldc + n;
                if (nv == 4 && ((((size_t)cp) & 7) == 0)) {
                    *reinterpret_cast<uint2*>(cp) = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (r < nv) cp[r] = f2bf(v[r]);
                }
            }
        }
    }
}

template <int BM, int BN, int AK, int BKD>
static void launch_variant(const toist_gemm& d, hipStream_t st) {
    dim3 grid((d.M + BM - 1) / BM, (d.N + BN - 1) / BN, d.batch * d.split_k);
    if (d.flags & 1)
        hipLaunchKernelGGL((gemm_kernel<BM, BN, AK, BKD, false>), grid, dim3(256), 0, st, d);
    else
        hipLaunchKernelGGL((gemm_kernel<BM, BN, AK, BKD, true>), grid, dim3(256), 0, st, d);
}

template <int BM, int BN>
static int launch_tile(const toist_gemm& d, hipStream_t st) {
    const int ak = d.a_kind, bk = d.b_kind;
    if (ak == TOIST_A_ROWK && bk == TOIST_B_ROWK) launch_variant<BM, BN, TOIST_A_ROWK, TOIST_B_ROWK>(d, st);
    else if (ak == TOIST_A_ROWK && bk == TOIST_B_KROW) launch_variant<BM, BN, TOIST_A_ROWK, TOIST_B_KROW>(d, st);
    else if (ak == TOIST_A_CONV && bk == TOIST_B_ROWK) launch_variant<BM, BN, TOIST_A_CONV, TOIST_B_ROWK>(d, st);
    else if (ak == TOIST_A_CONVT && bk == TOIST_B_KROW) launch_variant<BM, BN, TOIST_A_CONVT, TOIST_B_KROW>(d, st);
    else if (ak == TOIST_A_CONVT && bk == TOIST_B_ROWK) launch_variant<BM, BN, TOIST_A_CONVT, TOIST_B_ROWK>(d, st);
    else if (ak == TOIST_A_KROW && bk == TOIST_B_KROW) launch_variant<BM, BN, TOIST_A_KROW, TOIST_B_KROW>(d, st);
    else if (ak == TOIST_A_KROW && bk == TOIST_B_CONVX) launch_variant<BM, BN, TOIST_A_KROW, TOIST_B_CONVX>(d, st);
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
    if (d.a_kind == TOIST_A_CONVT) TOIST_REQUIRE((d.a.SC % 32) == 0, "toist_gemm_bf16: CONVT needs source channels %% 32 == 0");
    if (d.b_kind == TOIST_B_KROW && d.b.kin > 0) TOIST_REQUIRE((d.b.kin % 32) == 0, "toist_gemm_bf16: kin %% 32 != 0");
    if (d.b_kind == TOIST_B_CONVX) TOIST_REQUIRE((d.N % 8) == 0, "toist_gemm_bf16: CONVX needs N %% 8 == 0");
    // k-major operands are read in 8-row chunks: rows beyond M/N inside the last chunk are read
    // (and discarded), so ld must cover the rounded-up extent.
    if (d.a_kind == TOIST_A_KROW) TOIST_REQUIRE(d.a.ld >= ((d.M + 7) & ~7), "toist_gemm_bf16: A_KROW needs lda >= roundup8(M)");
    if (d.b_kind == TOIST_B_KROW) TOIST_REQUIRE(d.b.ld >= ((d.N + 7) & ~7), "toist_gemm_bf16: B_KROW needs ldb >= roundup8(N)");
    if (d.split_k > 1 || d.epi.accumulate)
        TOIST_REQUIRE(d.epi.out_f32, "toist_gemm_bf16: split_k/accumulate needs an f32 output");
    if (d.split_k > 1)
        TOIST_REQUIRE(!d.epi.shift && !d.epi.res && d.epi.act == TOIST_ACT_NONE && !d.epi.pre_out && d.epi.drop_where == 0,
                      "toist_gemm_bf16: split_k only supports alpha/scale epilogues");
    if (d.epi.act >= TOIST_ACT_MASK_POS) TOIST_REQUIRE(d.epi.aux != nullptr, "toist_gemm_bf16: activation %d needs aux", d.epi.act);
    if (d.epi.drop_where) TOIST_REQUIRE(d.epi.drop_p >= 0.f && d.epi.drop_p < 1.f, "toist_gemm_bf16: bad dropout p");

    int tile = d.tile;
    if (tile == 0) {
        const long long t128 = (long long)((d.M + 127) / 128) * ((d.N + 127) / 128) * d.batch * d.split_k;
        tile = (t128 >= 192 && d.M >= 128 && d.N >= 128) ? 128 : 64;
    }
    TOIST_REQUIRE(tile == 64 || tile == 128, "toist_gemm_bf16: tile must be 0, 64 or 128");
    int rc = (tile == 128) ? launch_tile<128, 128>(d, (hipStream_t)stream) : launch_tile<64, 64>(d, (hipStream_t)stream);
    if (rc != TOIST_OK) return rc;
    return check_launch("toist_gemm_bf16");
}
