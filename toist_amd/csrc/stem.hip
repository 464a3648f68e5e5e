// ResNet stem in ONE launch: conv 7x7 / stride 2 / pad 3 (3 -> 64 channels, FrozenBatchNorm folded: scale in the weights, shift here) + ReLU +
// max-pool 3x3 / stride 2 / pad 1, from the fp32 NCHW image batch to the bf16 NHWC input of layer 1 (torchvision ResNet.conv1 / bn1 / relu /
// maxpool, reached through /root/reference/models/backbone.py:64-91; the stem is frozen there, so nothing of it is needed by a backward pass).
//
// As three launches (pack to NHWC8, implicit GEMM on 64 x 64 tiles, max-pool) the stem moved 52 + 157 + 131 MB at batch 8 / 640 x 640 and took
// 22 + 155 + 31 us: the 7x7 gather stages a 416-deep k-tile in 16-byte pieces (one tap of one pixel each) for 64 output channels, and the
// 105 MB convolution output exists only to be pooled.  Here a workgroup owns an 8 x 8 tile of POOLED pixels:
//   * its 39 x 39 input patch is read from the three fp32 planes once, converted and laid out [pixel][8 channels] bf16 in LDS (24 KB);
//   * the 17 x 17 convolution outputs under the tile are 19 MFMA row blocks x 4 column blocks x 13 k-steps (k = 49 taps x 8 channels, 4 taps
//     per 32-deep step): the A fragment of a lane is ONE 16-byte LDS read at patch[(2 cy + r) * 39 + 2 cx + s], the weights sit in LDS as
//     ready-made B fragments (53 KB, loaded once per workgroup: the grid is persistent);
//   * shift + ReLU, bf16, into a [289][64] LDS tile (37 KB); the 3 x 3 / stride 2 maximum of every pooled pixel is nine 16-byte LDS reads.
// Convolution positions outside the image count as 0 in the maximum: every window holds at least one real position and all real values
// are >= 0 after the ReLU, so this equals the -inf padding of nn.MaxPool2d.  HBM traffic: 39 MB in, 26 MB out.
// Measured (batch 8, 640 x 640): 215 us as three launches, 120 us fused, 89 us with the next tile's patch in flight during the MFMAs.  Phase
// stamps: ~8.4k cycles of a 16.5k-cycle tile are the MFMA phase, bound by LDS reads (728 KB of fragment reads per tile: every wave re-reads
// the weights).  Tried and NOT kept: weights in registers with 5 x 2 wave tiles (255 VGPRs, 109 us), the same with the weights in LDS and
// fragments read one k-step ahead (121 us), an XOR swizzle of the patch slots (the stride-2 walk is conflict-free as laid out: a
// ds_read_b128 lane group mixes two k groups, whose walks fall on even and odd slots).
#include "common.h"

namespace toist {

constexpr int ST_PT = 8;                       // pooled tile edge
constexpr int ST_CT = 2 * ST_PT + 1;           // 17: convolution rows / columns under it
constexpr int ST_IT = 2 * ST_CT + 5;           // 39: input rows / columns under those
constexpr int ST_NPIX = ST_CT * ST_CT;         // 289
constexpr int ST_MF = (ST_NPIX + 15) / 16;     // 19 row blocks
constexpr int ST_KS = 13;                      // k-steps of 32 (49 taps x 8 channels = 392, padded to 416)
constexpr int ST_WB_BYTES = ST_KS * 4 * 64 * 16;         // 53 248
constexpr int ST_PATCH_BYTES = ST_IT * ST_IT * 16;       // 24 336
constexpr int ST_CP = 72;                      // elements per pixel of the convolution tile in LDS: 64 channels + 8 of padding (144 bytes).  With 128-byte pixels the
                                               // 16 lanes of a fragment column wrote 8 bytes each at a 128-byte stride: 16 of the 64 banks, an 8-way conflict on all 12
                                               // stores per lane and tile (63 % of the LDS-active cycles, profiles/r05_pmc_lds_conflicts.txt); 144 bytes spread them over all banks
constexpr int ST_CONV_BYTES = ST_NPIX * ST_CP * 2;       // 41 616
constexpr int ST_LDS = ST_WB_BYTES + ST_PATCH_BYTES + ST_CONV_BYTES;

__global__ __launch_bounds__(512) void stem_kernel(const float* __restrict__ img, const bf16_t* __restrict__ w, const float* __restrict__ shift,
                                                   bf16_t* __restrict__ out, int N, int C, int H, int W, int OH, int OW, int PH, int PW) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    uint4* const sW = reinterpret_cast<uint4*>(lds);                                   // [ks][nb][lane] B fragments
    bf16_t* const sP = reinterpret_cast<bf16_t*>(lds + ST_WB_BYTES);                   // [39*39][8]
    bf16_t* const sC = reinterpret_cast<bf16_t*>(lds + ST_WB_BYTES + ST_PATCH_BYTES);  // [289][ST_CP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c16 = lane & 15;

    // ---- once per workgroup: weights as B fragments (lane (column c16, k group g) of step ks = tap 4 ks + g, its 8 channels), zeroed patch ----
    for (int i = tid; i < ST_KS * 4 * 64; i += 512) {
        const int ln = i & 63, nb = (i >> 6) & 3, ks = i >> 8;
        const int tap = ks * 4 + (ln >> 4), n = nb * 16 + (ln & 15);
        sW[i] = tap < 49 ? *reinterpret_cast<const uint4*>(w + ((size_t)n * 49 + tap) * 8) : make_uint4(0u, 0u, 0u, 0u);
    }
    for (int i = tid; i < ST_IT * ST_IT; i += 512) reinterpret_cast<uint4*>(sP)[i] = make_uint4(0u, 0u, 0u, 0u);   // channels C .. 7 stay zero
    int toff[ST_KS];                            // byte offset of this lane's tap inside the patch, per k-step
#pragma unroll
    for (int ks = 0; ks < ST_KS; ++ks) {
        int tap = ks * 4 + g;
        if (tap > 48) tap = 48;                 // padding taps: their weights are zero, any valid address will do
        const int r = tap / 7, s_ = tap - r * 7;
        toff[ks] = (r * ST_IT + s_) * 16;
    }
    float sh[4][4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int j = 0; j < 4; ++j) sh[nb][j] = shift ? shift[nb * 16 + g * 4 + j] : 0.f;
    // the row blocks of this wave: wave, wave + 8, wave + 16 (19 in all); lane = convolution pixel 16 i + c16 of the tile
    int abase[3];
    bool plive[3];
    int pidx[3];
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int p = (wave + 8 * u) * 16 + c16;
        plive[u] = (wave + 8 * u) < ST_MF && p < ST_NPIX;
        pidx[u] = p;
        const int pc = p < ST_NPIX ? p : ST_NPIX - 1;
        const int cy = pc / ST_CT, cx = pc - cy * ST_CT;
        abase[u] = ((2 * cy) * ST_IT + 2 * cx) * 16;
    }
    const int tiles_x = (PW + ST_PT - 1) / ST_PT, tiles_y = (PH + ST_PT - 1) / ST_PT;
    const int tiles = tiles_x * tiles_y * N;
    __syncthreads();

    // The patch of the NEXT tile is requested (into registers) before this tile's MFMAs and written to LDS once they are done: alone on
    // its CU (114 KB of LDS), a workgroup would otherwise sit out a full HBM round trip per tile (120 us for the launch instead of ~50).
    constexpr int PL = (3 * ST_IT * ST_IT + 511) / 512;          // loads per thread (C <= 3: the usual case; more channels take the slow path below)
    float pre[PL];
    int p_r[PL], p_x[PL], p_c[PL], p_l[PL];     // patch row / column / plane / LDS element of this thread's loads (tile-invariant)
#pragma unroll
    for (int u = 0; u < PL; ++u) {
        const int i = tid + 512 * u;
        const int c = i / (ST_IT * ST_IT), rem = i - c * (ST_IT * ST_IT);
        p_c[u] = c;
        p_r[u] = rem / ST_IT;
        p_x[u] = rem - p_r[u] * ST_IT;
        p_l[u] = (c < C && c < 3) ? rem * 8 + c : -1;
    }
    auto fetch = [&](const int tile_) {
        const int n_ = tile_ / (tiles_x * tiles_y), trem_ = tile_ - n_ * (tiles_x * tiles_y);
        const int ty_ = trem_ / tiles_x, tx_ = trem_ - ty_ * tiles_x;
        const int iy0_ = 4 * ST_PT * ty_ - 5, ix0_ = 4 * ST_PT * tx_ - 5;
        const bool live = tile_ < tiles;
#pragma unroll
        for (int u = 0; u < PL; ++u) {
            const int iy = iy0_ + p_r[u], ix = ix0_ + p_x[u];
            float v = 0.f;
            if (live && p_l[u] >= 0 && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) v = img[(((size_t)n_ * C + p_c[u]) * H + iy) * W + ix];
            pre[u] = v;
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int u = 0; u < PL; ++u)
            if (p_l[u] >= 0) sP[p_l[u]] = f2bf(pre[u]);
    };
    const bool fast = C <= 3;
    if (fast) {
        fetch(blockIdx.x);
        stash();
    }
    for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int n = tile / (tiles_x * tiles_y), trem = tile - n * (tiles_x * tiles_y);
        const int ty = trem / tiles_x, tx = trem - ty * tiles_x;
        const int iy0 = 4 * ST_PT * ty - 5, ix0 = 4 * ST_PT * tx - 5;       // input origin of the patch
        const int cy0 = 2 * ST_PT * ty - 1, cx0 = 2 * ST_PT * tx - 1;       // convolution origin of the tile
        if (!fast) {    // ---- input patch: C fp32 planes -> bf16 [pixel][8] ----
            for (int i = tid; i < C * ST_IT * ST_IT; i += 512) {
                const int c = i / (ST_IT * ST_IT), rem = i - c * (ST_IT * ST_IT);
                const int r = rem / ST_IT, x = rem - r * ST_IT;
                const int iy = iy0 + r, ix = ix0 + x;
                float v = 0.f;
                if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) v = img[(((size_t)n * C + c) * H + iy) * W + ix];
                sP[rem * 8 + c] = f2bf(v);
            }
        }
        __syncthreads();
        if (fast) fetch(tile + (int)gridDim.x);                            // lands while the MFMAs below run

        // ---- convolution: k-steps outside, this wave's row blocks inside (the B fragments of a step are read once per wave) ----
        f32x4_t acc[3][4];
#pragma unroll
        for (int u = 0; u < 3; ++u)
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) acc[u][nb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        const bool third = wave + 16 < ST_MF;   // waves 0 .. 2 own three row blocks
#pragma unroll
        for (int ks = 0; ks < ST_KS; ++ks) {
            bf16x8_t bfr[4];
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) bfr[nb] = *reinterpret_cast<const bf16x8_t*>(&sW[(ks * 4 + nb) * 64 + lane]);
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                if (u == 2 && !third) continue;
                const bf16x8_t af = *reinterpret_cast<const bf16x8_t*>(reinterpret_cast<const unsigned char*>(sP) + abase[u] + toff[ks]);
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) acc[u][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[nb], af, acc[u][nb], 0, 0, 0);
            }
        }
        // shift + ReLU -> bf16 [pixel][64]; positions outside the convolution output plane count as 0
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            if (!plive[u]) continue;
            const int p = pidx[u];
            const int cy = p / ST_CT, cx = p - cy * ST_CT;
            const bool inside = (unsigned)(cy0 + cy) < (unsigned)OH && (unsigned)(cx0 + cx) < (unsigned)OW;
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = inside ? fmaxf(acc[u][nb][j] + sh[nb][j], 0.f) : 0.f;
                *reinterpret_cast<uint2*>(sC + p * ST_CP + nb * 16 + g * 4) = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
            }
        }
        __syncthreads();
        if (fast) stash();                      // every wave is done reading this tile's patch

        // ---- 3x3 / stride 2 maximum: thread = (pooled pixel, 8 channels) ----
        {
            const int q = tid >> 3, c8 = tid & 7, py = q >> 3, px = q & 7;
            const int gy = ST_PT * ty + py, gx = ST_PT * tx + px;
            if (gy < PH && gx < PW) {
                uint4 m = make_uint4(0u, 0u, 0u, 0u);                  // all candidates are >= 0: packed signed 16-bit maxima order bf16 correctly
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        const uint4 v = *reinterpret_cast<const uint4*>(sC + ((2 * py + dy) * ST_CT + 2 * px + dx) * ST_CP + c8 * 8);
                        asm("v_pk_max_i16 %0, %0, %1" : "+v"(m.x) : "v"(v.x));
                        asm("v_pk_max_i16 %0, %0, %1" : "+v"(m.y) : "v"(v.y));
                        asm("v_pk_max_i16 %0, %0, %1" : "+v"(m.z) : "v"(v.z));
                        asm("v_pk_max_i16 %0, %0, %1" : "+v"(m.w) : "v"(v.w));
                    }
                *reinterpret_cast<uint4*>(out + (((size_t)n * PH + gy) * PW + gx) * 64 + c8 * 8) = m;
            }
        }
        __syncthreads();                        // the patch and the convolution tile are free again
    }
}

}  // namespace toist

using namespace toist;

extern "C" int toist_stem_fwd(const float* image, const void* weight, const float* shift, int N, int C, int H, int W, void* out, void* stream) {
    TOIST_REQUIRE(image != nullptr && weight != nullptr && out != nullptr, "toist_stem_fwd: null pointer");
    TOIST_REQUIRE(N > 0 && C > 0 && C <= 8 && H > 0 && W > 0, "toist_stem_fwd: bad shape N=%d C=%d H=%d W=%d (C <= 8)", N, C, H, W);
    TOIST_REQUIRE((((size_t)weight) & 15) == 0 && (((size_t)out) & 15) == 0, "toist_stem_fwd: weight / out must be 16-byte aligned");
    const int OH = (H + 6 - 7) / 2 + 1, OW = (W + 6 - 7) / 2 + 1;
    const int PH = (OH + 2 - 3) / 2 + 1, PW = (OW + 2 - 3) / 2 + 1;
    static std::atomic<unsigned long long> done{0};
    if (!lds_attr_once_flag(done, [] { return hipFuncSetAttribute((const void*)stem_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, ST_LDS) == hipSuccess; })) {
        set_last_error("toist_stem_fwd: cannot enable %d bytes of LDS", ST_LDS);
        return TOIST_EHIP;
    }
    const long long tiles = (long long)((PH + ST_PT - 1) / ST_PT) * ((PW + ST_PT - 1) / ST_PT) * N;
    const int grid = (int)(tiles < 256 ? tiles : 256);          // persistent: one workgroup per CU keeps the weight fragments in LDS
    hipLaunchKernelGGL(stem_kernel, dim3(grid), dim3(512), ST_LDS, (hipStream_t)stream, image, (const bf16_t*)weight, shift, (bf16_t*)out, N, C, H, W,
                       OH, OW, PH, PW);
    return check_launch("toist_stem_fwd");
}
