// Segmentation-branch kernels (config 3): /root/reference/models/segmentation.py
//   attnmap_softmax  : the per-head softmax over H x W of MHAttentionMap.forward (:262-273, `flatten(3)`)
//   groupnorm(+ReLU) : torch.nn.GroupNorm(8, C) + F.relu of MaskHeadSmallConv.forward (:203-241), NHWC
//   upsample_add     : `cur_fpn + F.interpolate(x, size, mode="nearest")` with the FPN term shared by all queries
//   sum_over_queries : reduction used by the backward of every "expand(...)" broadcast (:204-205)
//   mask_loss        : bilinear upsample + sigmoid focal + dice of SetCriterion.loss_masks (mdetr.py:827-853,
//                      segmentation.py:276-319), forward and backward
// All of these are HBM-bound passes over bf16 activations (16-byte vector accesses, fp32 math).
#include "common.h"

namespace toist {

__device__ __forceinline__ void unpack8(const uint4& u, float* f) {
    f[0] = __uint_as_float(u.x << 16); f[1] = __uint_as_float(u.x & 0xffff0000u);
    f[2] = __uint_as_float(u.y << 16); f[3] = __uint_as_float(u.y & 0xffff0000u);
    f[4] = __uint_as_float(u.z << 16); f[5] = __uint_as_float(u.z & 0xffff0000u);
    f[6] = __uint_as_float(u.w << 16); f[7] = __uint_as_float(u.w & 0xffff0000u);
}
__device__ __forceinline__ uint4 pack8(const float* f) {
    return make_uint4(pack2bf(f[0], f[1]), pack2bf(f[2], f[3]), pack2bf(f[4], f[5]), pack2bf(f[6], f[7]));
}

__device__ __forceinline__ float block_reduce_sum(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}
__device__ __forceinline__ float block_reduce_max(float v, float* red) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// ------------------------------------------------------------------------------- attention-map softmax
// MHAttentionMap.forward normalises `weights.flatten(3)`, i.e. over the H*W positions of EACH head
// (segmentation.py:271).  scores [B,Q,H,ld] bf16 (ld >= HW) -> probabilities channels-last [B*Q, HW, H] bf16
// (heads innermost: the layout the mask head's first convolution gathers).  One wave per (b,q,head) row.
__global__ __launch_bounds__(256) void attnmap_softmax_fwd_kernel(const bf16_t* __restrict__ s, const unsigned char* __restrict__ key_pad,
                                                                   int rows, int Q, int H, int HW, int ld, bf16_t* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);   // (b*Q + q)*H + h
    if (row >= rows) return;
    const int bq = row / H, h = row - bq * H, b = bq / Q;
    const bf16_t* sr = s + (size_t)row * ld;
    float mx = -INFINITY;
    for (int p = lane; p < HW; p += 64) {
        const float v = (key_pad && key_pad[(size_t)b * HW + p]) ? -INFINITY : bf2f(sr[p]);
        mx = fmaxf(mx, v);
    }
    mx = wave_max(mx);
    float sum = 0.f;
    for (int p = lane; p < HW; p += 64) {
        const float v = (key_pad && key_pad[(size_t)b * HW + p]) ? -INFINITY : bf2f(sr[p]);
        sum += __expf(v - mx);
    }
    sum = wave_sum(sum);
    const float inv = 1.f / sum;
    for (int p = lane; p < HW; p += 64) {
        const float v = (key_pad && key_pad[(size_t)b * HW + p]) ? -INFINITY : bf2f(sr[p]);
        out[((size_t)bq * HW + p) * H + h] = f2bf(__expf(v - mx) * inv);
    }
}
// dscores[b,q,h,p] = P * (dP - sum_p(P*dP)) per head; prob / dprob are channels-last [B*Q, HW, H]
__global__ __launch_bounds__(256) void attnmap_softmax_bwd_kernel(const bf16_t* __restrict__ prob, const bf16_t* __restrict__ dprob, int rows, int H,
                                                                   int HW, int ld, bf16_t* __restrict__ ds) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int bq = row / H, h = row - bq * H;
    float dot = 0.f;
    for (int p = lane; p < HW; p += 64) {
        const size_t j = ((size_t)bq * HW + p) * H + h;
        dot += bf2f(prob[j]) * bf2f(dprob[j]);
    }
    dot = wave_sum(dot);
    for (int p = lane; p < ld; p += 64) {
        float v = 0.f;
        if (p < HW) {
            const size_t j = ((size_t)bq * HW + p) * H + h;
            v = bf2f(prob[j]) * (bf2f(dprob[j]) - dot);
        }
        ds[(size_t)row * ld + p] = f2bf(v);
    }
}

// ------------------------------------------------------------------------------- GroupNorm (+ReLU), NHWC
// Work split shared by the reducing GroupNorm kernels: a block owns a slab of pixels of sample n; thread t always
// handles channel chunk (t % c8) of pixel (t / c8) + k * (256 / c8), so its 8 channels -- and their groups -- are fixed
// and every partial sum stays in registers until one LDS/global atomic per channel at the end.
// stats[n][g] = {sum, sum of squares} over the (HW x C/G) elements of group g of sample n (f32 atomics).
__global__ __launch_bounds__(256) void gn_stats_kernel(const bf16_t* __restrict__ x, int HW, int C, int G, float* __restrict__ stats) {
    __shared__ float acc[2][16];
    const int n = blockIdx.y;
    const int Cg = C / G, c8 = C >> 3;
    if (threadIdx.x < 32) acc[threadIdx.x >> 4][threadIdx.x & 15] = 0.f;
    __syncthreads();
    const int ppi = 256 / c8;                      // pixels per block iteration
    const int cc = threadIdx.x % c8, pl = threadIdx.x / c8;
    const int per = (HW + gridDim.x - 1) / gridDim.x;
    const int p_beg = blockIdx.x * per, p_end = (p_beg + per < HW) ? p_beg + per : HW;
    float s[8], q[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s[j] = 0.f; q[j] = 0.f; }
    if (pl < ppi) {
        // four pixels per trip: the loads of a trip are issued together (one 16-byte load in flight per thread ran this pass at 3 TB/s)
        int p = p_beg + pl;
        for (; p + 3 * ppi < p_end; p += 4 * ppi) {
            uint4 r[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) r[u] = *reinterpret_cast<const uint4*>(x + ((size_t)n * HW + p + u * ppi) * C + cc * 8);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float v[8];
                unpack8(r[u], v);
#pragma unroll
                for (int j = 0; j < 8; ++j) { s[j] += v[j]; q[j] += v[j] * v[j]; }
            }
        }
        for (; p < p_end; p += ppi) {
            float v[8];
            unpack8(*reinterpret_cast<const uint4*>(x + ((size_t)n * HW + p) * C + cc * 8), v);
#pragma unroll
            for (int j = 0; j < 8; ++j) { s[j] += v[j]; q[j] += v[j] * v[j]; }
        }
        // lanes c8 apart hold the same channels: fold them inside the wave first (with C = 16 the 32 lanes per chunk used to queue on ONE LDS
        // address per atomic -- 32 atomics x 32-way x 22 blocks per CU were ~150 of this pass's 215 us at 160 x 160)
        const bool pow2 = (c8 & (c8 - 1)) == 0 && c8 < 64;
        if (pow2) {
            for (int m = c8; m < 64; m <<= 1) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { s[j] += __shfl_xor(s[j], m); q[j] += __shfl_xor(q[j], m); }
            }
        }
        if (!pow2 || (int)(threadIdx.x & 63) < c8) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int g = (cc * 8 + j) / Cg;
                atomicAdd(&acc[0][g], s[j]);
                atomicAdd(&acc[1][g], q[j]);
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < G) {
        atomicAdd(stats + ((size_t)n * G + threadIdx.x) * 2, acc[0][threadIdx.x]);
        atomicAdd(stats + ((size_t)n * G + threadIdx.x) * 2 + 1, acc[1][threadIdx.x]);
    }
}
// y = relu((x - mean) * rstd * gamma + beta).  Work split of the reducing kernels above (round 5): a thread keeps ONE channel chunk and walks the
// pixels of its block's slab, so the affine coefficients y = x * a + b of its 8 channels (two statistics loads, a division and an rsqrt each) are
// computed once -- a thread per 16-byte chunk spent most of its instructions on them (the 20 x 20 and 40 x 40 stages ran at 1.2-1.6 TB/s).
__global__ __launch_bounds__(256) void gn_apply_kernel(const bf16_t* __restrict__ x, const float* __restrict__ stats, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, int HW, int C, int G, float eps, int relu,
                                                        bf16_t* __restrict__ y) {
    const int n = blockIdx.y;
    const int Cg = C / G, c8 = C >> 3;
    const float cnt = (float)HW * (float)Cg;
    const int ppi = 256 / c8;
    const int cc = threadIdx.x % c8, pl = threadIdx.x / c8;
    if (pl >= ppi) return;
    const int per = (HW + gridDim.x - 1) / gridDim.x;
    const int p_beg = blockIdx.x * per, p_end = (p_beg + per < HW) ? p_beg + per : HW;
    float ca[8], cb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = cc * 8 + j, g = c / Cg;
        const float mean = stats[((size_t)n * G + g) * 2] / cnt;
        const float var = fmaxf(stats[((size_t)n * G + g) * 2 + 1] / cnt - mean * mean, 0.f);
        ca[j] = rsqrtf(var + eps) * gamma[c];
        cb[j] = beta[c] - mean * ca[j];
    }
    for (int p = p_beg + pl; p < p_end; p += ppi) {
        float v[8];
        const size_t off = ((size_t)n * HW + p) * C + cc * 8;
        unpack8(*reinterpret_cast<const uint4*>(x + off), v);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float o = v[j] * ca[j] + cb[j];
            v[j] = relu ? fmaxf(o, 0.f) : o;
        }
        *reinterpret_cast<uint4*>(y + off) = pack8(v);
    }
}
// backward pass 1: with g = dy * (y > 0): bstats[n][g] = {sum g*gamma, sum g*gamma*xhat}; dgamma[c] += sum g*xhat; dbeta[c] += sum g
__global__ __launch_bounds__(256) void gn_bwd_stats_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ y, const bf16_t* __restrict__ x,
                                                            const float* __restrict__ stats, const float* __restrict__ gamma, const float* __restrict__ beta, int HW, int C, int G,
                                                            float eps, int relu, float* __restrict__ bstats, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta) {
    extern __shared__ float sm[];  // [2][C] channel partials + [2][16] group partials
    float* cg = sm;
    float* cb = sm + C;
    float* ga = sm + 2 * C;
    const int n = blockIdx.y;
    const int Cg = C / G, c8 = C >> 3;
    const float cnt = (float)HW * (float)Cg;
    for (int i = threadIdx.x; i < 2 * C + 32; i += 256) sm[i] = 0.f;
    __syncthreads();
    const int ppi = 256 / c8;
    const int cc = threadIdx.x % c8, pl = threadIdx.x / c8;
    const int per = (HW + gridDim.x - 1) / gridDim.x;
    const int p_beg = blockIdx.x * per, p_end = (p_beg + per < HW) ? p_beg + per : HW;
    if (pl < ppi) {
        // y == nullptr (round 5): the ReLU mask is re-derived from x with the forward's own arithmetic (gn_apply_kernel: o = x * a + b) instead of
        // reading the output tensor a second and third time -- two of the seven passes over the activation that the two backward kernels made
        const bool remask = relu && y == nullptr;
        float mean[8], rs[8], gm[8], a_gx[8], a_g[8], fa[8], fb[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = cc * 8 + j, g = c / Cg;
            mean[j] = stats[((size_t)n * G + g) * 2] / cnt;
            const float var = fmaxf(stats[((size_t)n * G + g) * 2 + 1] / cnt - mean[j] * mean[j], 0.f);
            rs[j] = rsqrtf(var + eps);
            gm[j] = gamma[c];
            fa[j] = rs[j] * gm[j];
            fb[j] = remask ? beta[c] - mean[j] * fa[j] : 0.f;
            a_gx[j] = 0.f; a_g[j] = 0.f;
        }
        auto one = [&](const uint4 rd, const uint4 rx, const size_t off) {
            float d[8], yy[8], xv[8];
            unpack8(rd, d);
            unpack8(rx, xv);
            if (remask) {
#pragma unroll
                for (int j = 0; j < 8; ++j) yy[j] = xv[j] * fa[j] + fb[j];
            } else if (relu) unpack8(*reinterpret_cast<const uint4*>(y + off), yy);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float gj = (relu && !(yy[j] > 0.f)) ? 0.f : d[j];
                a_gx[j] += gj * (xv[j] - mean[j]) * rs[j];
                a_g[j] += gj;
            }
        };
        int p = p_beg + pl;
        for (; p + ppi < p_end; p += 2 * ppi) {          // two pixels per trip: four loads in flight per thread
            const size_t o0 = ((size_t)n * HW + p) * C + cc * 8, o1 = o0 + (size_t)ppi * C;
            const uint4 d0 = *reinterpret_cast<const uint4*>(dy + o0), x0 = *reinterpret_cast<const uint4*>(x + o0);
            const uint4 d1 = *reinterpret_cast<const uint4*>(dy + o1), x1 = *reinterpret_cast<const uint4*>(x + o1);
            one(d0, x0, o0);
            one(d1, x1, o1);
        }
        for (; p < p_end; p += ppi) {
            const size_t o0 = ((size_t)n * HW + p) * C + cc * 8;
            one(*reinterpret_cast<const uint4*>(dy + o0), *reinterpret_cast<const uint4*>(x + o0), o0);
        }
        const bool pow2 = (c8 & (c8 - 1)) == 0 && c8 < 64;      // see gn_stats_kernel
        if (pow2) {
            for (int m = c8; m < 64; m <<= 1) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { a_gx[j] += __shfl_xor(a_gx[j], m); a_g[j] += __shfl_xor(a_g[j], m); }
            }
        }
        if (!pow2 || (int)(threadIdx.x & 63) < c8) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int c = cc * 8 + j, g = c / Cg;
                atomicAdd(&cg[c], a_gx[j]);
                atomicAdd(&cb[c], a_g[j]);
                atomicAdd(&ga[g], a_g[j] * gm[j]);
                atomicAdd(&ga[16 + g], a_gx[j] * gm[j]);
            }
        }
    }
    __syncthreads();
    if (dgamma)
        for (int c = threadIdx.x; c < C; c += 256) { atomicAdd(dgamma + c, cg[c]); atomicAdd(dbeta + c, cb[c]); }
    if (threadIdx.x < G) {
        atomicAdd(bstats + ((size_t)n * G + threadIdx.x) * 2, ga[threadIdx.x]);
        atomicAdd(bstats + ((size_t)n * G + threadIdx.x) * 2 + 1, ga[16 + threadIdx.x]);
    }
}
// backward pass 2: dx = rstd * (g*gamma - mean_g(g*gamma) - xhat * mean_g(g*gamma*xhat)); thread = one channel chunk, as in gn_apply_kernel
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ y, const bf16_t* __restrict__ x,
                                                            const float* __restrict__ stats, const float* __restrict__ bstats,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta, int HW, int C, int G, float eps, int relu,
                                                            bf16_t* __restrict__ dx) {
    const int n = blockIdx.y;
    const int Cg = C / G, c8 = C >> 3;
    const float cnt = (float)HW * (float)Cg;
    const int ppi = 256 / c8;
    const int cc = threadIdx.x % c8, pl = threadIdx.x / c8;
    if (pl >= ppi) return;
    const int per = (HW + gridDim.x - 1) / gridDim.x;
    const int p_beg = blockIdx.x * per, p_end = (p_beg + per < HW) ? p_beg + per : HW;
    const bool remask = relu && y == nullptr;       // see gn_bwd_stats_kernel
    float mu[8], rsd[8], gm[8], m1[8], m2[8], fa[8], fb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = cc * 8 + j, g = c / Cg;
        mu[j] = stats[((size_t)n * G + g) * 2] / cnt;
        const float var = fmaxf(stats[((size_t)n * G + g) * 2 + 1] / cnt - mu[j] * mu[j], 0.f);
        rsd[j] = rsqrtf(var + eps);
        gm[j] = gamma[c];
        fa[j] = rsd[j] * gm[j];
        fb[j] = remask ? beta[c] - mu[j] * fa[j] : 0.f;
        m1[j] = bstats[((size_t)n * G + g) * 2] / cnt;
        m2[j] = bstats[((size_t)n * G + g) * 2 + 1] / cnt;
    }
    for (int p = p_beg + pl; p < p_end; p += ppi) {
        float d[8], yy[8], xv[8];
        const size_t off = ((size_t)n * HW + p) * C + cc * 8;
        unpack8(*reinterpret_cast<const uint4*>(dy + off), d);
        unpack8(*reinterpret_cast<const uint4*>(x + off), xv);
        if (remask) {
#pragma unroll
            for (int j = 0; j < 8; ++j) yy[j] = xv[j] * fa[j] + fb[j];
        } else if (relu) unpack8(*reinterpret_cast<const uint4*>(y + off), yy);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float xh = (xv[j] - mu[j]) * rsd[j];
            const float gj = (relu && !(yy[j] > 0.f)) ? 0.f : d[j];
            d[j] = rsd[j] * (gj * gm[j] - m1[j] - xh * m2[j]);
        }
        *reinterpret_cast<uint4*>(dx + off) = pack8(d);
    }
}
// ------------------------------------------------------------------------------- nearest resize + shared FPN term
// out[bq, Y, X, :] = fpn[bq / Q, Y, X, :] + in[bq, src(Y), src(X), :]   (in is [BQ, H, W, C], out [BQ, OH, OW, C]) with F.interpolate(mode="nearest")'s
// source index min(floor(dst * (float)in / out), in - 1) (segmentation.py:218, 225, 232: the maps are resized to the FPN level's own size, which is
// 2H x 2W only when the image sides are multiples of 32).
// rows != nullptr: map i of `in` / `out` is map rows[i] of the batch (a gathered subset), its image rows[i] / Q
__device__ __forceinline__ int nearest_src(int dst, float scale, int in) {
    const int s = (int)floorf((float)dst * scale);
    return s < in - 1 ? s : in - 1;
}
template <bool twice>        // twice: OH = 2 H and OW = 2 W (image sides that are multiples of 32: the shift form, no float index arithmetic)
__global__ __launch_bounds__(256) void upsample_add_kernel(const bf16_t* __restrict__ in, const bf16_t* __restrict__ fpn, const long long* __restrict__ rows,
                                                            int BQ, int Q, int H, int W, int OH, int OW, int C, bf16_t* __restrict__ out) {
    const int c8 = C >> 3;
    const float sy = (float)H / (float)OH, sx = (float)W / (float)OW;
    const long long total = (long long)BQ * OH * OW * c8;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int cc = (int)(i % c8);
        long long p = i / c8;
        const int X = (int)(p % OW); p /= OW;
        const int Y = (int)(p % OH);
        const int bq = (int)(p / OH);
        const int ys = twice ? (Y >> 1) : nearest_src(Y, sy, H), xs = twice ? (X >> 1) : nearest_src(X, sx, W);
        float a[8], f[8];
        unpack8(*reinterpret_cast<const uint4*>(in + ((((size_t)bq * H + ys) * W + xs) * C) + cc * 8), a);
        const int img = (int)((rows != nullptr ? rows[bq] : (long long)bq) / Q);
        unpack8(*reinterpret_cast<const uint4*>(fpn + ((((size_t)img * OH + Y) * OW + X) * C) + cc * 8), f);
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] += f[j];
        *reinterpret_cast<uint4*>(out + (size_t)i * 8) = pack8(a);
    }
}
// din[bq, y, x, :] = sum of the output pixels that read it (the transpose of the map above: 4 of them for the exact doubling)
template <bool twice>
__global__ __launch_bounds__(256) void upsample_add_bwd_kernel(const bf16_t* __restrict__ dout, int BQ, int H, int W, int OH, int OW, int C, bf16_t* __restrict__ din) {
    const int c8 = C >> 3;
    const float sy = (float)H / (float)OH, sx = (float)W / (float)OW;
    const long long total = (long long)BQ * H * W * c8;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int cc = (int)(i % c8);
        long long p = i / c8;
        const int x = (int)(p % W); p /= W;
        const int y = (int)(p % H);
        const int bq = (int)(p / H);
        float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int Y0, Y1, X0, X1;
        if (twice) {
            Y0 = 2 * y; Y1 = 2 * y + 1; X0 = 2 * x; X1 = 2 * x + 1;
        } else {       // candidates: a window that certainly holds every Y with src(Y) == y; membership is tested with the forward map itself
            Y0 = max(0, (int)((float)y / sy) - 2); Y1 = min(OH - 1, (int)((float)(y + 1) / sy) + 2);
            X0 = max(0, (int)((float)x / sx) - 2); X1 = min(OW - 1, (int)((float)(x + 1) / sx) + 2);
        }
        for (int Y = Y0; Y <= Y1; ++Y) {
            if (!twice && nearest_src(Y, sy, H) != y) continue;
            for (int X = X0; X <= X1; ++X) {
                if (!twice && nearest_src(X, sx, W) != x) continue;
                float v[8];
                unpack8(*reinterpret_cast<const uint4*>(dout + ((((size_t)bq * OH + Y) * OW + X) * C) + cc * 8), v);
#pragma unroll
                for (int j = 0; j < 8; ++j) a[j] += v[j];
            }
        }
        *reinterpret_cast<uint4*>(din + (size_t)i * 8) = pack8(a);
    }
}
// out[b, i] = sum_q in[b, q, i]   (i over HW*C elements, 8 per thread)
__global__ __launch_bounds__(256) void sum_queries_kernel(const bf16_t* __restrict__ in, int B, int Q, long long per8, bf16_t* __restrict__ out) {
    const long long total = (long long)B * per8;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long b = i / per8, r = i - b * per8;
        float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int q = 0; q < Q; ++q) {
            float v[8];
            unpack8(reinterpret_cast<const uint4*>(in)[(b * Q + q) * per8 + r], v);
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] += v[j];
        }
        reinterpret_cast<uint4*>(out)[i] = pack8(a);
    }
}

// out[b, i] = sum over the maps s in [seg[b], seg[b+1]) of in[s, i]: the per-image reduction when only a subset of the maps (the matched
// queries, packed image by image) carries a gradient.  seg is read on the device (the hipGraph replay serves any batch).
__global__ __launch_bounds__(256) void sum_segments_kernel(const bf16_t* __restrict__ in, const int* __restrict__ seg, int B, int rows, long long per8,
                                                            bf16_t* __restrict__ out) {
    const long long total = (long long)B * per8;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long b = i / per8, r = i - b * per8;
        int s0 = seg[b], s1 = seg[b + 1];
        if (s0 < 0) s0 = 0;
        if (s1 > rows) s1 = rows;
        float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int s = s0; s < s1; ++s) {
            float v[8];
            unpack8(reinterpret_cast<const uint4*>(in)[(long long)s * per8 + r], v);
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] += v[j];
        }
        reinterpret_cast<uint4*>(out)[i] = pack8(a);
    }
}

// ------------------------------------------------------------------------------- mask losses
// For matched pair t: prediction map pred[pred_row[t]] [h,w] f32 is bilinearly upsampled (align_corners=False) to
// [TH,TW] and compared with gt[gt_row[t]] (u8 [TH,TW], zero padded like NestedTensor.from_tensor_list).  Accumulates per pair
// sums[t] = {sum focal, sum p*t, sum p, sum t}.
__device__ __forceinline__ void bilinear_src(int o, int in, int out, int& i0, int& i1, float& w1) {
    float s = ((float)o + 0.5f) * ((float)in / (float)out) - 0.5f;
    if (s < 0.f) s = 0.f;
    i0 = (int)s;
    i1 = (i0 < in - 1) ? i0 + 1 : i0;
    w1 = s - (float)i0;
}
__global__ __launch_bounds__(256) void mask_loss_fwd_kernel(const float* __restrict__ pred, const int* __restrict__ pred_row,
                                                             const unsigned char* __restrict__ gt, const int* __restrict__ gt_row,
                                                             int h, int w, int TH, int TW, float alpha, float* __restrict__ sums,
                                                             const int* __restrict__ valid_hw) {
    __shared__ float red[4];
    const int t = blockIdx.y;
    if (pred_row[t] < 0) return;          // an unused slot of a fixed-capacity pair table (matcher.StaticTargets): its sums stay 0 -> both losses 0
    // valid_hw (device int32 [4], optional) = {VH, VW, hs, ws}: the batch's own padded size inside a larger [TH, TW] bucket (harness.CapturedTrainStep) and
    // the part of the prediction the reference would have had for that batch.  The reference resizes its [hs, ws] prediction to the batch's largest
    // image [VH, VW] (mdetr.py:843): here prediction rows / columns [0, hs) x [0, ws) are mapped onto target pixels [0, VH) x [0, VW) with the
    // same align_corners=False grid, and nothing else takes part.
    const int VH = valid_hw ? min(valid_hw[0], TH) : TH, VW = valid_hw ? min(valid_hw[1], TW) : TW;
    const int hs = valid_hw ? min(valid_hw[2], h) : h, ws = valid_hw ? min(valid_hw[3], w) : w;
    const float* pm = pred + (size_t)pred_row[t] * h * w;
    const unsigned char* gm = gt + (size_t)gt_row[t] * TH * TW;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    const int total = TH * TW;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int Y = i / TW, X = i - Y * TW;
        if (Y >= VH || X >= VW) continue;
        int y0, y1, x0, x1; float wy, wx;
        bilinear_src(Y, hs, VH, y0, y1, wy);
        bilinear_src(X, ws, VW, x0, x1, wx);
        const float v = (1.f - wy) * ((1.f - wx) * pm[y0 * w + x0] + wx * pm[y0 * w + x1]) + wy * ((1.f - wx) * pm[y1 * w + x0] + wx * pm[y1 * w + x1]);
        const float tg = gm[i] ? 1.f : 0.f;
        const float p = 1.f / (1.f + __expf(-v));
        const float ce = fmaxf(v, 0.f) - v * tg + log1pf(__expf(-fabsf(v)));   // BCE with logits
        const float pt = p * tg + (1.f - p) * (1.f - tg);
        const float at = alpha * tg + (1.f - alpha) * (1.f - tg);
        a0 += at * ce * (1.f - pt) * (1.f - pt);
        a1 += p * tg; a2 += p; a3 += tg;
    }
    a0 = block_reduce_sum(a0, red); a1 = block_reduce_sum(a1, red); a2 = block_reduce_sum(a2, red); a3 = block_reduce_sum(a3, red);
    if (threadIdx.x == 0) {
        atomicAdd(sums + t * 4 + 0, a0); atomicAdd(sums + t * 4 + 1, a1); atomicAdd(sums + t * 4 + 2, a2); atomicAdd(sums + t * 4 + 3, a3);
    }
}
// d(loss_mask*g_f + loss_dice*g_d)/d pred, scattered back through the bilinear weights (f32 atomics into dpred,
// which the caller zeroes).  scale_f = g_f / (TH*TW*num_boxes), scale_d = g_d / num_boxes are read from `coef`.
// Tiled form: a workgroup owns an MLB_TILE x MLB_TILE block of target pixels of one pair, whose bilinear footprint in the prediction is
// at most MLB_SRC x MLB_SRC source pixels; contributions are summed in LDS and only the footprint goes out as global
// atomics (the plain form issued four global atomics per target pixel: 65 M per step).
constexpr int MLB_TILE = 64, MLB_SRC = 36, MLB_CAND = 16;   // 64 x 64 target pixels per workgroup (16 per thread: the LDS window's zeroing and write-back were most of a 32 x 32 tile's time); the window fits up-sampling ratios >= 1.9, others take the global-atomic path
__global__ __launch_bounds__(256) void mask_loss_bwd_kernel(const float* __restrict__ pred, const int* __restrict__ pred_row,
                                                             const unsigned char* __restrict__ gt, const int* __restrict__ gt_row,
                                                             int h, int w, int TH, int TW, float alpha, const float* __restrict__ sums,
                                                             const float* __restrict__ coef, float* __restrict__ dpred, int compact,
                                                             const int* __restrict__ valid_hw) {
    __shared__ float acc[MLB_SRC * MLB_SRC];
    __shared__ float win[MLB_SRC * MLB_SRC];
    __shared__ float gvs[MLB_TILE * MLB_TILE];
    const int t = blockIdx.y;
    if (pred_row[t] < 0) return;          // unused slot
    const float* pm = pred + (size_t)pred_row[t] * h * w;
    float* dp = dpred + (size_t)(compact ? t : pred_row[t]) * h * w;   // compact: the gradient of pair t goes to row t of a [T,h,w] buffer
    const unsigned char* gm = gt + (size_t)gt_row[t] * TH * TW;
    const float sf = coef[0], sd = coef[1];
    const float num = 2.f * sums[t * 4 + 1] + 1.f, den = sums[t * 4 + 2] + sums[t * 4 + 3] + 1.f;
    const int tiles_x = (TW + MLB_TILE - 1) / MLB_TILE;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int Y0 = ty * MLB_TILE, X0 = tx * MLB_TILE;
    const int VH = valid_hw ? min(valid_hw[0], TH) : TH, VW = valid_hw ? min(valid_hw[1], TW) : TW;      // as in the forward kernel
    const int hs = valid_hw ? min(valid_hw[2], h) : h, ws = valid_hw ? min(valid_hw[3], w) : w;
    if (Y0 >= VH || X0 >= VW) return;
    // source origin of this tile (first source row / column any of its pixels touches)
    int sy0, sx0, dummy; float wdummy;
    bilinear_src(Y0, hs, VH, sy0, dummy, wdummy);
    bilinear_src(X0, ws, VW, sx0, dummy, wdummy);
    // Up-sampling ratios in [1.9, 5.5] (the reference predicts masks at 1/4 of the padded image): the tile's footprint fits the LDS window and a
    // source pixel is touched by at most MLB_CAND target rows / columns -> GATHER: per-target-pixel gradients go to LDS with plain stores and
    // every source pixel of the window sums its own contributions (the scatter form below spent 420 of its 520 us in same-address LDS float
    // atomics: measured 110 us with the atomics replaced by stores).  Other ratios keep the scatter form.
    const float sc_y = (float)VH / (float)hs, sc_x = (float)VW / (float)ws;
    const bool gather = sc_y >= 1.9f && sc_y <= 5.5f && sc_x >= 1.9f && sc_x <= 5.5f;     // 2 * 5.5 + 5 = MLB_CAND candidate columns
    // the prediction window under this tile goes to LDS once
    for (int i = threadIdx.x; i < MLB_SRC * MLB_SRC; i += 256) {
        acc[i] = 0.f;
        const int y = sy0 + i / MLB_SRC, x = sx0 + i % MLB_SRC;
        win[i] = (y < hs && x < ws) ? pm[y * w + x] : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < MLB_TILE * MLB_TILE; i += 256) {
        const int Y = Y0 + i / MLB_TILE, X = X0 + i % MLB_TILE;
        if (Y >= VH || X >= VW) {
            gvs[i] = 0.f;
            continue;
        }
        int y0, y1, x0, x1; float wy, wx;
        bilinear_src(Y, hs, VH, y0, y1, wy);
        bilinear_src(X, ws, VW, x0, x1, wx);
        const int ly0 = y0 - sy0, ly1 = y1 - sy0, lx0 = x0 - sx0, lx1 = x1 - sx0;
        const bool inside = ly1 < MLB_SRC && lx1 < MLB_SRC;      // always true in gather mode
        float p00, p01, p10, p11;
        if (inside) {
            p00 = win[ly0 * MLB_SRC + lx0]; p01 = win[ly0 * MLB_SRC + lx1]; p10 = win[ly1 * MLB_SRC + lx0]; p11 = win[ly1 * MLB_SRC + lx1];
        } else {
            p00 = pm[y0 * w + x0]; p01 = pm[y0 * w + x1]; p10 = pm[y1 * w + x0]; p11 = pm[y1 * w + x1];
        }
        const float v = (1.f - wy) * ((1.f - wx) * p00 + wx * p01) + wy * ((1.f - wx) * p10 + wx * p11);
        const float tg = gm[(size_t)Y * TW + X] ? 1.f : 0.f;
        const float p = 1.f / (1.f + __expf(-v));
        const float ce = fmaxf(v, 0.f) - v * tg + log1pf(__expf(-fabsf(v)));
        const float pt = p * tg + (1.f - p) * (1.f - tg);
        const float at = alpha * tg + (1.f - alpha) * (1.f - tg);
        // focal: d/dv [at * ce * (1-pt)^2], dce/dv = p - t, dpt/dv = (2t-1) p (1-p)
        const float dfocal = at * ((p - tg) * (1.f - pt) * (1.f - pt) - ce * 2.f * (1.f - pt) * (2.f * tg - 1.f) * p * (1.f - p));
        // dice: loss = 1 - num/den ; d/dp = -(2 t den - num) / den^2
        const float ddice = -(2.f * tg * den - num) / (den * den) * p * (1.f - p);
        const float gv = sf * dfocal + sd * ddice;
        if (gather) {
            gvs[i] = gv;
        } else if (inside) {
            atomicAdd(&acc[ly0 * MLB_SRC + lx0], gv * (1.f - wy) * (1.f - wx));
            atomicAdd(&acc[ly0 * MLB_SRC + lx1], gv * (1.f - wy) * wx);
            atomicAdd(&acc[ly1 * MLB_SRC + lx0], gv * wy * (1.f - wx));
            atomicAdd(&acc[ly1 * MLB_SRC + lx1], gv * wy * wx);
        } else {      // a footprint larger than the LDS window (down-sampling ratios): global atomics
            atomicAdd(dp + y0 * w + x0, gv * (1.f - wy) * (1.f - wx));
            atomicAdd(dp + y0 * w + x1, gv * (1.f - wy) * wx);
            atomicAdd(dp + y1 * w + x0, gv * wy * (1.f - wx));
            atomicAdd(dp + y1 * w + x1, gv * wy * wx);
        }
    }
    __syncthreads();
    if (!gather) {
        for (int i = threadIdx.x; i < MLB_SRC * MLB_SRC; i += 256) {
            const float a = acc[i];
            const int y = sy0 + i / MLB_SRC, x = sx0 + i % MLB_SRC;
            if (a != 0.f && y < hs && x < ws) atomicAdd(dp + y * w + x, a);
        }
        return;
    }
    // gather: source pixel (gy, gx) of the window receives sum over the tile's target pixels of gv * Wy(Y -> gy) * Wx(X -> gx), where a target row
    // Y gives weight (1 - wy) to its upper source row y0 and wy to y1 (both to the same row at the image border) -- the scatter form's weights
    const int Yend = min(Y0 + MLB_TILE, VH) - 1, Xend = min(X0 + MLB_TILE, VW) - 1;
    for (int s = threadIdx.x; s < MLB_SRC * MLB_SRC; s += 256) {
        const int gy = sy0 + s / MLB_SRC, gx = sx0 + s % MLB_SRC;
        if (gy >= hs || gx >= ws) continue;
        // targets whose bilinear support can contain this source pixel: (o + 0.5) / sc - 0.5 in [g - 1, g + 1), widened by one each side
        const int Ylo = max(Y0, (int)floorf(((float)gy - 0.5f) * sc_y - 0.5f) - 1), Yhi = min(Yend, (int)ceilf(((float)gy + 1.5f) * sc_y - 0.5f) + 1);
        const int Xlo = max(X0, (int)floorf(((float)gx - 0.5f) * sc_x - 0.5f) - 1), Xhi = min(Xend, (int)ceilf(((float)gx + 1.5f) * sc_x - 0.5f) + 1);
        if (Ylo > Yhi || Xlo > Xhi) continue;
        float wxv[MLB_CAND];
#pragma unroll
        for (int j = 0; j < MLB_CAND; ++j) {
            int x0, x1; float wx;
            bilinear_src(Xlo + j, ws, VW, x0, x1, wx);
            wxv[j] = (Xlo + j <= Xhi) ? ((x0 == gx ? 1.f - wx : 0.f) + (x1 == gx ? wx : 0.f)) : 0.f;
        }
        float total = 0.f;
        for (int Y = Ylo; Y <= Yhi; ++Y) {
            int y0, y1; float wy;
            bilinear_src(Y, hs, VH, y0, y1, wy);
            const float wyk = (y0 == gy ? 1.f - wy : 0.f) + (y1 == gy ? wy : 0.f);
            if (wyk == 0.f) continue;
            const float* row = gvs + (Y - Y0) * MLB_TILE + (Xlo - X0);
            float rs = 0.f;
#pragma unroll
            for (int j = 0; j < MLB_CAND; ++j)
                if (Xlo + j <= Xhi) rs += wxv[j] * row[j];
            total += wyk * rs;
        }
        if (total != 0.f) atomicAdd(dp + gy * w + gx, total);
    }
}

static inline int grid_cap(long long n, int cap = 4096) {
    long long g = (n + 255) / 256;
    if (g < 1) g = 1;
    return (int)(g > cap ? cap : g);
}

// pixel slabs per sample for the GroupNorm apply kernels: the reducing kernels' split, widened when few samples would leave CUs idle
static inline int gn_apply_slabs(int gx, int N, int HW, int C) {
    const int ppi = 256 / (C / 8);
    if ((long long)N * gx < 1024) {
        int want = (1024 + N - 1) / N, most = HW / (ppi * 4);
        if (most < 1) most = 1;
        if (want > most) want = most;
        if (want > gx) gx = want;
    }
    return gx;
}

}  // namespace toist

using namespace toist;

extern "C" int toist_attnmap_softmax_fwd(const void* scores, const uint8_t* key_pad, int B, int Q, int H, int HW, int ld, void* out, void* stream) {
    TOIST_REQUIRE(B > 0 && Q > 0 && H > 0 && HW > 0 && ld >= HW, "toist_attnmap_softmax_fwd: bad shape");
    const int rows = B * Q * H;
    hipLaunchKernelGGL(attnmap_softmax_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)scores, key_pad, rows, Q, H,
                       HW, ld, (bf16_t*)out);
    return check_launch("toist_attnmap_softmax_fwd");
}
extern "C" int toist_attnmap_softmax_bwd(const void* prob, const void* dprob, int BQ, int H, int HW, int ld, void* dscores, void* stream) {
    TOIST_REQUIRE(BQ > 0 && H > 0 && HW > 0 && ld >= HW, "toist_attnmap_softmax_bwd: bad shape");
    const int rows = BQ * H;
    hipLaunchKernelGGL(attnmap_softmax_bwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)prob, (const bf16_t*)dprob, rows,
                       H, HW, ld, (bf16_t*)dscores);
    return check_launch("toist_attnmap_softmax_bwd");
}

extern "C" int toist_groupnorm_fwd(const void* x, const float* gamma, const float* beta, int N, int HW, int C, int G, float eps, int relu,
                                   void* y, float* stats, void* stream) {
    TOIST_REQUIRE(N > 0 && HW > 0 && C > 0 && (C % 8) == 0 && C / 8 <= 256 && G > 0 && G <= 16 && (C % G) == 0, "toist_groupnorm_fwd: bad shape (C%%8==0, C<=2048, G<=16)");
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(stats, 0, sizeof(float) * 2 * (size_t)N * G, st);
    if (e != hipSuccess) { set_last_error("toist_groupnorm_fwd: memset: %s", hipGetErrorString(e)); return TOIST_EHIP; }
    const long long total = (long long)HW * (C / 8);
    int gx = (int)((total + 8191) / 8192);   // pixel slabs per sample
    if (gx > 32) gx = 32;
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL(gn_stats_kernel, dim3(gx, N), dim3(256), 0, st, (const bf16_t*)x, HW, C, G, stats);
    if (y == nullptr) return check_launch("toist_groupnorm_fwd");      // statistics only: the consumer normalises on the way in (toist_mask_stage_fwd)
    hipLaunchKernelGGL(gn_apply_kernel, dim3(gn_apply_slabs(gx, N, HW, C), N), dim3(256), 0, st, (const bf16_t*)x, stats, gamma, beta, HW, C, G, eps, relu,
                       (bf16_t*)y);
    return check_launch("toist_groupnorm_fwd");
}
// y = [relu](GroupNorm(x)) from statistics computed earlier (toist_groupnorm_fwd / toist_mask_stage_fwd): the backward of the fused mask stages
// re-creates the normalised activation of the maps it needs instead of keeping it for all of them
extern "C" int toist_groupnorm_apply(const void* x, const float* stats, const float* gamma, const float* beta, int N, int HW, int C, int G, float eps, int relu,
                                     void* y, void* stream) {
    TOIST_REQUIRE(x && stats && gamma && beta && y && N > 0 && HW > 0 && C > 0 && (C % 8) == 0 && C / 8 <= 256 && G > 0 && G <= 16 && (C % G) == 0,
                  "toist_groupnorm_apply: bad shape (C%%8==0, C<=2048, G<=16)");
    const long long total = (long long)HW * (C / 8);
    int gx = (int)((total + 8191) / 8192);
    if (gx > 32) gx = 32;
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL(gn_apply_kernel, dim3(gn_apply_slabs(gx, N, HW, C), N), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, stats, gamma, beta, HW, C, G,
                       eps, relu, (bf16_t*)y);
    return check_launch("toist_groupnorm_apply");
}
extern "C" int toist_groupnorm_bwd(const void* dy, const void* y, const void* x, const float* stats, const float* gamma, const float* beta, int N, int HW,
                                   int C, int G, float eps, int relu, void* dx, float* dgamma, float* dbeta, float* bstats, void* stream) {
    TOIST_REQUIRE(N > 0 && HW > 0 && C > 0 && (C % 8) == 0 && G > 0 && G <= 16 && (C % G) == 0 && C / 8 <= 256, "toist_groupnorm_bwd: bad shape");
    TOIST_REQUIRE(!relu || y != nullptr || beta != nullptr, "toist_groupnorm_bwd: the ReLU mask needs y, or beta to re-derive it from x");
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(bstats, 0, sizeof(float) * 2 * (size_t)N * G, st);
    if (e != hipSuccess) { set_last_error("toist_groupnorm_bwd: memset: %s", hipGetErrorString(e)); return TOIST_EHIP; }
    const long long total = (long long)HW * (C / 8);
    int gx = (int)((total + 8191) / 8192);
    if (gx > 32) gx = 32;
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL(gn_bwd_stats_kernel, dim3(gx, N), dim3(256), sizeof(float) * (2 * C + 32), st, (const bf16_t*)dy, (const bf16_t*)y,
                       (const bf16_t*)x, stats, gamma, beta, HW, C, G, eps, relu, bstats, dgamma, dbeta);
    hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3(gn_apply_slabs(gx, N, HW, C), N), dim3(256), 0, st, (const bf16_t*)dy, (const bf16_t*)y, (const bf16_t*)x, stats,
                       bstats, gamma, beta, HW, C, G, eps, relu, (bf16_t*)dx);
    return check_launch("toist_groupnorm_bwd");
}

static int launch_resize_add(const char* what, const void* in, const void* fpn, const int64_t* rows, int n, int Q, int H, int W, int OH, int OW, int C, void* out, void* stream) {
    TOIST_REQUIRE(in && fpn && out && n > 0 && Q > 0 && H > 0 && W > 0 && OH >= H && OW >= W && (C % 8) == 0 && (rows || (n % Q) == 0), "%s: bad shape", what);
    if (OH == 2 * H && OW == 2 * W)
        hipLaunchKernelGGL(upsample_add_kernel<true>, dim3(grid_cap((long long)n * OH * OW * (C / 8), 8192)), dim3(256), 0, (hipStream_t)stream,
                           (const bf16_t*)in, (const bf16_t*)fpn, (const long long*)rows, n, Q, H, W, OH, OW, C, (bf16_t*)out);
    else
        hipLaunchKernelGGL(upsample_add_kernel<false>, dim3(grid_cap((long long)n * OH * OW * (C / 8), 8192)), dim3(256), 0, (hipStream_t)stream,
                           (const bf16_t*)in, (const bf16_t*)fpn, (const long long*)rows, n, Q, H, W, OH, OW, C, (bf16_t*)out);
    return check_launch(what);
}
extern "C" int toist_upsample_add(const void* in, const void* fpn, int BQ, int Q, int H, int W, int C, void* out, void* stream) {
    return launch_resize_add("toist_upsample_add", in, fpn, nullptr, BQ, Q, H, W, 2 * H, 2 * W, C, out, stream);
}
extern "C" int toist_upsample_add_rows(const void* in, const void* fpn, const int64_t* rows, int n, int Q, int H, int W, int C, void* out, void* stream) {
    TOIST_REQUIRE(rows != nullptr, "toist_upsample_add_rows: rows is required");
    return launch_resize_add("toist_upsample_add_rows", in, fpn, rows, n, Q, H, W, 2 * H, 2 * W, C, out, stream);
}
extern "C" int toist_resize_add(const void* in, const void* fpn, const int64_t* rows, int n, int Q, int H, int W, int OH, int OW, int C, void* out, void* stream) {
    return launch_resize_add("toist_resize_add", in, fpn, rows, n, Q, H, W, OH, OW, C, out, stream);
}
static int launch_resize_add_bwd(const char* what, const void* dout, int BQ, int H, int W, int OH, int OW, int C, void* din, void* stream) {
    TOIST_REQUIRE(dout && din && BQ > 0 && H > 0 && W > 0 && OH >= H && OW >= W && (C % 8) == 0, "%s: bad shape", what);
    if (OH == 2 * H && OW == 2 * W)
        hipLaunchKernelGGL(upsample_add_bwd_kernel<true>, dim3(grid_cap((long long)BQ * H * W * (C / 8), 8192)), dim3(256), 0, (hipStream_t)stream,
                           (const bf16_t*)dout, BQ, H, W, OH, OW, C, (bf16_t*)din);
    else
        hipLaunchKernelGGL(upsample_add_bwd_kernel<false>, dim3(grid_cap((long long)BQ * H * W * (C / 8), 8192)), dim3(256), 0, (hipStream_t)stream,
                           (const bf16_t*)dout, BQ, H, W, OH, OW, C, (bf16_t*)din);
    return check_launch(what);
}
extern "C" int toist_upsample_add_bwd(const void* dout, int BQ, int H, int W, int C, void* din, void* stream) {
    return launch_resize_add_bwd("toist_upsample_add_bwd", dout, BQ, H, W, 2 * H, 2 * W, C, din, stream);
}
extern "C" int toist_resize_add_bwd(const void* dout, int BQ, int H, int W, int OH, int OW, int C, void* din, void* stream) {
    return launch_resize_add_bwd("toist_resize_add_bwd", dout, BQ, H, W, OH, OW, C, din, stream);
}
extern "C" int toist_sum_queries(const void* in, int B, int Q, int64_t per, void* out, void* stream) {
    TOIST_REQUIRE(B > 0 && Q > 0 && per > 0 && (per % 8) == 0, "toist_sum_queries: bad shape");
    hipLaunchKernelGGL(sum_queries_kernel, dim3(grid_cap((long long)B * (per / 8), 8192)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)in, B, Q,
                       (long long)(per / 8), (bf16_t*)out);
    return check_launch("toist_sum_queries");
}
extern "C" int toist_sum_segments(const void* in, const int32_t* seg, int B, int rows, int64_t per, void* out, void* stream) {
    TOIST_REQUIRE(in && seg && out && B > 0 && rows > 0 && per > 0 && (per % 8) == 0, "toist_sum_segments: bad shape");
    hipLaunchKernelGGL(sum_segments_kernel, dim3(grid_cap((long long)B * (per / 8), 8192)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)in, seg, B, rows,
                       (long long)(per / 8), (bf16_t*)out);
    return check_launch("toist_sum_segments");
}

extern "C" int toist_mask_loss_fwd(const float* pred, const int32_t* pred_row, const uint8_t* gt, const int32_t* gt_row, int T, int h, int w,
                                   int TH, int TW, float alpha, float* sums, const int32_t* valid_hw, void* stream) {
    TOIST_REQUIRE(T > 0 && h > 0 && w > 0 && TH > 0 && TW > 0, "toist_mask_loss_fwd: bad shape");
    hipLaunchKernelGGL(mask_loss_fwd_kernel, dim3(grid_cap((long long)TH * TW, 64), T), dim3(256), 0, (hipStream_t)stream, pred, pred_row, gt, gt_row, h, w,
                       TH, TW, alpha, sums, valid_hw);
    return check_launch("toist_mask_loss_fwd");
}
extern "C" int toist_mask_loss_bwd(const float* pred, const int32_t* pred_row, const uint8_t* gt, const int32_t* gt_row, int T, int h, int w,
                                   int TH, int TW, float alpha, const float* sums, const float* coef, float* dpred, const int32_t* valid_hw, void* stream) {
    TOIST_REQUIRE(T > 0 && h > 0 && w > 0 && TH > 0 && TW > 0, "toist_mask_loss_bwd: bad shape");
    hipLaunchKernelGGL(mask_loss_bwd_kernel, dim3(((TH + MLB_TILE - 1) / MLB_TILE) * ((TW + MLB_TILE - 1) / MLB_TILE), T), dim3(256), 0, (hipStream_t)stream, pred, pred_row, gt, gt_row, h, w,
                       TH, TW, alpha, sums, coef, dpred, 0, valid_hw);
    return check_launch("toist_mask_loss_bwd");
}
extern "C" int toist_mask_loss_bwd_compact(const float* pred, const int32_t* pred_row, const uint8_t* gt, const int32_t* gt_row, int T, int h, int w,
                                           int TH, int TW, float alpha, const float* sums, const float* coef, float* dpred_rows, const int32_t* valid_hw, void* stream) {
    TOIST_REQUIRE(T > 0 && h > 0 && w > 0 && TH > 0 && TW > 0, "toist_mask_loss_bwd_compact: bad shape");
    hipLaunchKernelGGL(mask_loss_bwd_kernel, dim3(((TH + MLB_TILE - 1) / MLB_TILE) * ((TW + MLB_TILE - 1) / MLB_TILE), T), dim3(256), 0, (hipStream_t)stream, pred, pred_row, gt, gt_row, h, w,
                       TH, TW, alpha, sums, coef, dpred_rows, 1, valid_hw);
    return check_launch("toist_mask_loss_bwd_compact");
}
