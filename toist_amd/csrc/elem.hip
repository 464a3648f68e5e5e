// Backbone-side elementwise kernels (HBM-bound, NHWC bf16):
//   image pack  : fp32 NCHW [N,3,H,W] -> bf16 NHWC with C padded to 8 (stem implicit-GEMM operand)
//   maxpool     : the 3x3 / stride-2 / pad-1 max pool of the ResNet stem (torchvision resnet.py, used
//                 through /root/reference/models/backbone.py:87-89)
//   unpack      : bf16 NHWC -> fp32 NCHW (API-edge feature maps, e.g. memory_cache / mask head inputs)
#include "common.h"

namespace toist {

__global__ __launch_bounds__(256) void pack_nchw_kernel(const float* __restrict__ in, int N, int C, int H, int W,
                                                         bf16_t* __restrict__ out) {
    const long long npix = (long long)N * H * W;
    for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < npix; p += (long long)gridDim.x * 256) {
        const long long n = p / ((long long)H * W);
        const long long hw = p - n * H * W;
        float v[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] = (c < C) ? in[(n * C + c) * H * W + hw] : 0.f;
        reinterpret_cast<uint4*>(out)[p] = make_uint4(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7]));
    }
}

__global__ __launch_bounds__(256) void maxpool_kernel(const bf16_t* __restrict__ in, int N, int H, int W, int C, int OH, int OW,
                                                       bf16_t* __restrict__ out) {
    const int c8 = C >> 3;
    const long long total = (long long)N * OH * OW * c8;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int cc = (int)(i % c8);
        long long p = i / c8;
        const int ox = (int)(p % OW); p /= OW;
        const int oy = (int)(p % OH);
        const int n = (int)(p / OH);
        float m[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) m[j] = -INFINITY;
        for (int dy = 0; dy < 3; ++dy) {
            const int iy = oy * 2 - 1 + dy;
            if (iy < 0 || iy >= H) continue;
            for (int dx = 0; dx < 3; ++dx) {
                const int ix = ox * 2 - 1 + dx;
                if (ix < 0 || ix >= W) continue;
                const uint4 u = *reinterpret_cast<const uint4*>(in + (((long long)n * H + iy) * W + ix) * C + cc * 8);
                const unsigned w4[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    m[2 * j] = fmaxf(m[2 * j], __uint_as_float(w4[j] << 16));
                    m[2 * j + 1] = fmaxf(m[2 * j + 1], __uint_as_float(w4[j] & 0xffff0000u));
                }
            }
        }
        *reinterpret_cast<uint4*>(out + (((long long)n * OH + oy) * OW + ox) * C + cc * 8) =
            make_uint4(pack2bf(m[0], m[1]), pack2bf(m[2], m[3]), pack2bf(m[4], m[5]), pack2bf(m[6], m[7]));
    }
}

// bf16 NHWC -> f32 NCHW through a 32x32 LDS tile (coalesced on both sides)
__global__ __launch_bounds__(256) void unpack_nhwc_kernel(const bf16_t* __restrict__ in, int HW, int C, float* __restrict__ out) {
    __shared__ float tile[32][33];
    const int n = blockIdx.z;
    const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const int p = p0 + r, c = c0 + tx;
        tile[r][tx] = (p < HW && c < C) ? bf2f(in[((long long)n * HW + p) * C + c]) : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r, p = p0 + tx;
        if (c < C && p < HW) out[((long long)n * C + c) * HW + p] = tile[tx][r];
    }
}


// PositionEmbeddingSine (normalize=True, scale=2*pi): /root/reference/models/position_encoding.py:30-49.
// mask [B,H,W] u8 (1 = padding) -> pos bf16 [B, H*W, 2*F] (token-major, y half then x half) and/or
// f32 [B, 2*F, H, W] (the reference layout).  One thread per (b, y, x, feature pair).
__global__ __launch_bounds__(256) void sine_pos_kernel(const unsigned char* __restrict__ mask, int B, int H, int W, int F,
                                                        float temperature, bf16_t* __restrict__ out_tok, float* __restrict__ out_nchw, int S) {
    // out_tok holds S >= H * W rows per image: rows beyond the image's tokens (the caption's tokens behind them in the cross-modal
    // encoder's sequence) get a zero encoding (transformer.py:139)
    const int half = F >> 1;
    const long long total = (long long)B * H * W * half, tail = (long long)B * (S - H * W) * half;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total + tail; i += (long long)gridDim.x * 256) {
        if (i >= total) {
            const long long t = i - total;
            const int j = (int)(t % half);
            const long long r = t / half;
            const int b = (int)(r / (S - H * W)), row = H * W + (int)(r % (S - H * W));
            bf16_t* o = out_tok + ((long long)b * S + row) * (2 * F);
            o[2 * j] = 0; o[2 * j + 1] = 0; o[F + 2 * j] = 0; o[F + 2 * j + 1] = 0;
            continue;
        }
        const int j = (int)(i % half);
        long long p = i / half;
        const int x = (int)(p % W); p /= W;
        const int y = (int)(p % H);
        const int b = (int)(p / H);
        const unsigned char* m = mask + (long long)b * H * W;
        float ye = 0.f, yt = 0.f, xe = 0.f, xt = 0.f;
        for (int yy = 0; yy < H; ++yy) { const float v = m[yy * W + x] ? 0.f : 1.f; yt += v; if (yy <= y) ye += v; }
        for (int xx = 0; xx < W; ++xx) { const float v = m[y * W + xx] ? 0.f : 1.f; xt += v; if (xx <= x) xe += v; }
        const float two_pi = 6.283185307179586f;
        ye = ye / (yt + 1e-6f) * two_pi;
        xe = xe / (xt + 1e-6f) * two_pi;
        const float dim_t = powf(temperature, (2.f * (float)j) / (float)F);
        const float vy0 = sinf(ye / dim_t), vy1 = cosf(ye / dim_t);
        const float vx0 = sinf(xe / dim_t), vx1 = cosf(xe / dim_t);
        const long long tok = (long long)b * S + (long long)y * W + x;
        if (out_tok) {
            bf16_t* o = out_tok + tok * (2 * F);
            o[2 * j] = f2bf(vy0); o[2 * j + 1] = f2bf(vy1);
            o[F + 2 * j] = f2bf(vx0); o[F + 2 * j + 1] = f2bf(vx1);
        }
        if (out_nchw) {
            const long long hw = (long long)H * W, pix = (long long)y * W + x;
            float* o = out_nchw + (long long)b * 2 * F * hw;
            o[(2 * j) * hw + pix] = vy0; o[(2 * j + 1) * hw + pix] = vy1;
            o[(F + 2 * j) * hw + pix] = vx0; o[(F + 2 * j + 1) * hw + pix] = vx1;
        }
    }
}

}  // namespace toist

using namespace toist;

extern "C" int toist_pack_image(const float* nchw, int N, int C, int H, int W, void* nhwc8, void* stream) {
    TOIST_REQUIRE(N > 0 && C > 0 && C <= 8 && H > 0 && W > 0, "toist_pack_image: bad shape (C <= 8)");
    long long npix = (long long)N * H * W;
    int grid = (int)((npix + 255) / 256 > 4096 ? 4096 : (npix + 255) / 256);
    hipLaunchKernelGGL(pack_nchw_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, nchw, N, C, H, W, (bf16_t*)nhwc8);
    return check_launch("toist_pack_image");
}

extern "C" int toist_maxpool3x3s2(const void* in, int N, int H, int W, int C, void* out, void* stream) {
    TOIST_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && (C % 8) == 0, "toist_maxpool3x3s2: bad shape (C %% 8 == 0)");
    const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
    long long total = (long long)N * OH * OW * (C / 8);
    int grid = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipLaunchKernelGGL(maxpool_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)in, N, H, W, C, OH, OW, (bf16_t*)out);
    return check_launch("toist_maxpool3x3s2");
}

extern "C" int toist_unpack_nhwc(const void* nhwc, int N, int HW, int C, float* nchw, void* stream) {
    TOIST_REQUIRE(N > 0 && HW > 0 && C > 0, "toist_unpack_nhwc: bad shape");
    hipLaunchKernelGGL(unpack_nhwc_kernel, dim3((HW + 31) / 32, (C + 31) / 32, N), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)nhwc,
                       HW, C, nchw);
    return check_launch("toist_unpack_nhwc");
}

// RoBERTa's position ids and the key-padding bytes of a tokenized batch in one launch: position = cumsum(id != pad) * (id != pad) + pad
// (HF create_position_ids_from_input_ids, reached through /root/reference/models/transformer.py:129-133), key_pad = (attention_mask != 1).
__global__ __launch_bounds__(64) void text_prep_kernel(const long long* __restrict__ ids, const long long* __restrict__ att, int B, int L, long long pad_id,
                                                       long long* __restrict__ pos_ids, unsigned char* __restrict__ key_pad) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    long long run = 0;
    for (int t = 0; t < L; ++t) {
        const long long keep = ids[(size_t)b * L + t] != pad_id ? 1 : 0;
        run += keep;
        pos_ids[(size_t)b * L + t] = run * keep + pad_id;
        key_pad[(size_t)b * L + t] = att[(size_t)b * L + t] != 1 ? 1 : 0;
    }
}

extern "C" int toist_text_prep(const int64_t* ids, const int64_t* attention_mask, int B, int L, int64_t pad_id, int64_t* pos_ids, uint8_t* key_pad, void* stream) {
    TOIST_REQUIRE(ids && attention_mask && pos_ids && key_pad && B > 0 && L > 0, "toist_text_prep: bad args");
    hipLaunchKernelGGL(text_prep_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, (const long long*)ids, (const long long*)attention_mask, B, L,
                       (long long)pad_id, (long long*)pos_ids, key_pad);
    return check_launch("toist_text_prep");
}

// Diagnostic: the constant-rate device clock (100 MHz, s_memrealtime) at the moment this one-thread kernel runs, i.e. when everything
// ordered before it on its stream has finished.  Captured into the step's hipGraph it dates the branches of a replayed step without a
// profiler attached (bench.py --stamps).
__global__ void stamp_kernel(unsigned long long* __restrict__ slots, int idx) { slots[idx] = wall_clock64(); }

extern "C" int toist_stamp(uint64_t* slots, int idx, void* stream) {
    TOIST_REQUIRE(slots != nullptr && idx >= 0, "toist_stamp: bad args");
    hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (unsigned long long*)slots, idx);
    return check_launch("toist_stamp");
}

static int sine_position(const uint8_t* mask, int B, int H, int W, int num_pos_feats, float temperature, void* out_tok, float* out_nchw, int S,
                         void* stream, const char* who) {
    TOIST_REQUIRE(B > 0 && H > 0 && W > 0 && num_pos_feats > 0 && (num_pos_feats % 2) == 0 && S >= H * W && (S == H * W || out_tok != nullptr), "%s: bad shape", who);
    long long total = (long long)B * S * (num_pos_feats / 2);
    int grid = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipLaunchKernelGGL(sine_pos_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, mask, B, H, W, num_pos_feats, temperature,
                       (bf16_t*)out_tok, out_nchw, S);
    return check_launch(who);
}

extern "C" int toist_sine_position(const uint8_t* mask, int B, int H, int W, int num_pos_feats, float temperature, void* out_tok,
                                   float* out_nchw, void* stream) {
    return sine_position(mask, B, H, W, num_pos_feats, temperature, out_tok, out_nchw, H * W, stream, "toist_sine_position");
}

extern "C" int toist_sine_position_seq(const uint8_t* mask, int B, int H, int W, int num_pos_feats, float temperature, void* out_tok, int rows_per_image,
                                       void* stream) {
    return sine_position(mask, B, H, W, num_pos_feats, temperature, out_tok, nullptr, rows_per_image, stream, "toist_sine_position_seq");
}
