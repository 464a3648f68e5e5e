// Row-wise and elementwise kernels of the TOIST transformer path (HBM-bound; one wave per row,
// 16-byte vector accesses, fp32 statistics).
//   LayerNorm fwd/bwd      -> nn.LayerNorm in /root/reference/models/transformer.py:279-280,341-345,481
//   masked softmax fwd/bwd -> the softmax(+key_padding_mask, +dropout) inside nn.MultiheadAttention
//                             (transformer.py:273,337-338) and HF RobertaSelfAttention
//   column sum             -> bias gradients of nn.Linear
//   add / dropout          -> with_pos_embed (transformer.py:287-288,357-358) and nn.Dropout
#include "common.h"
#include <cstdlib>

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

// ---------------------------------------------------------------------------------- LayerNorm
constexpr int LN_MAXCH = 2;  // D <= 1024

__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const bf16_t* __restrict__ x, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float eps, int rows, int D,
                                                             bf16_t* __restrict__ y, float* __restrict__ mean_out,
                                                             float* __restrict__ rstd_out, const bf16_t* __restrict__ add,
                                                             bf16_t* __restrict__ y2) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nch = D >> 3;
    float v[LN_MAXCH][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXCH; ++i) {
        const int ch = lane + 64 * i;
        if (ch < nch) {
            unpack8(*reinterpret_cast<const uint4*>(x + (size_t)row * D + ch * 8), v[i]);
#pragma unroll
            for (int j = 0; j < 8; ++j) s += v[i][j];
        }
    }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXCH; ++i) {
        const int ch = lane + 64 * i;
        if (ch < nch) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float d = v[i][j] - mean; q += d * d; }
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
#pragma unroll
    for (int i = 0; i < LN_MAXCH; ++i) {
        const int ch = lane + 64 * i;
        if (ch < nch) {
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (v[i][j] - mean) * rstd * gamma[ch * 8 + j] + beta[ch * 8 + j];
            const uint4 yo = pack8(o);
            *reinterpret_cast<uint4*>(y + (size_t)row * D + ch * 8) = yo;
            if (y2 != nullptr) {   // second output y + add (the positional / query embedding the next attention adds to its q, k input):
                float a8[8], y8[8];   // bf16(bf16(y) + add), exactly what the separate add kernel computed from the stored y
                unpack8(*reinterpret_cast<const uint4*>(add + (size_t)row * D + ch * 8), a8);
                unpack8(yo, y8);
#pragma unroll
                for (int j = 0; j < 8; ++j) y8[j] += a8[j];
                *reinterpret_cast<uint4*>(y2 + (size_t)row * D + ch * 8) = pack8(y8);
            }
        }
    }
    if (lane == 0) {
        if (mean_out) mean_out[row] = mean;
        if (rstd_out) rstd_out[row] = rstd;
    }
}

// dx = rstd * (dy*g - mean(dy*g) - xhat * mean(dy*g*xhat)); dgamma += dy*xhat; dbeta += dy.
// Optional dx_drop = dropout-masked copy of dx (gradient entering a residual branch that was
// `x + dropout(branch)` in the forward pass; mask regenerated from (seed, element index)).
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                             const float* __restrict__ mean, const float* __restrict__ rstd,
                                                             const float* __restrict__ gamma, int rows, int D,
                                                             bf16_t* __restrict__ dx, float* __restrict__ dgamma,
                                                             float* __restrict__ dbeta, bf16_t* __restrict__ dx_drop,
                                                             float drop_p, unsigned long long seed,
                                                             const unsigned long long* __restrict__ seed_dev, float* __restrict__ partials) {
    if (seed_dev) seed += *seed_dev;
    const int lane = threadIdx.x & 63;
    const int wave_global = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * 4;
    const int nch = D >> 3;
    float ag[LN_MAXCH][8], ab[LN_MAXCH][8];
#pragma unroll
    for (int i = 0; i < LN_MAXCH; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) { ag[i][j] = 0.f; ab[i][j] = 0.f; }
    const unsigned thresh = dx_drop ? (unsigned)(drop_p * 4294967296.0) : 0u;
    const float dscale = dx_drop ? 1.f / (1.f - drop_p) : 1.f;

    // a wave walks its rows serially: the next row's operands are requested before the current row is reduced, and gamma stays in
    // registers (one exposed load round trip per wave instead of one per row)
    float gam[LN_MAXCH][8];
    uint4 nd[LN_MAXCH], nx[LN_MAXCH];
    float nmu = 0.f, nrs = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXCH; ++i) {
        const int ch = lane + 64 * i;
#pragma unroll
        for (int j = 0; j < 8; ++j) gam[i][j] = (ch < nch) ? gamma[ch * 8 + j] : 0.f;
        nd[i] = make_uint4(0, 0, 0, 0);
        nx[i] = make_uint4(0, 0, 0, 0);
    }
    auto fetch = [&](int r) {
#pragma unroll
        for (int i = 0; i < LN_MAXCH; ++i) {
            const int ch = lane + 64 * i;
            if (ch < nch) {
                nd[i] = *reinterpret_cast<const uint4*>(dy + (size_t)r * D + ch * 8);
                nx[i] = *reinterpret_cast<const uint4*>(x + (size_t)r * D + ch * 8);
            }
        }
        nmu = mean[r];
        nrs = rstd[r];
    };
    if (wave_global < rows) fetch(wave_global);
    for (int row = wave_global; row < rows; row += nwaves) {
        const float mu = nmu, rs = nrs;
        uint4 cd[LN_MAXCH], cx[LN_MAXCH];
#pragma unroll
        for (int i = 0; i < LN_MAXCH; ++i) { cd[i] = nd[i]; cx[i] = nx[i]; }
        if (row + nwaves < rows) fetch(row + nwaves);
        float g[LN_MAXCH][8], xh[LN_MAXCH][8];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXCH; ++i) {
            const int ch = lane + 64 * i;
            if (ch < nch) {
                float d[8], xv[8];
                unpack8(cd[i], d);
                unpack8(cx[i], xv);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    xh[i][j] = (xv[j] - mu) * rs;
                    g[i][j] = d[j] * gam[i][j];
                    s1 += g[i][j];
                    s2 += g[i][j] * xh[i][j];
                    ag[i][j] += d[j] * xh[i][j];
                    ab[i][j] += d[j];
                }
            }
        }
        s1 = wave_sum(s1) / (float)D;
        s2 = wave_sum(s2) / (float)D;
#pragma unroll
        for (int i = 0; i < LN_MAXCH; ++i) {
            const int ch = lane + 64 * i;
            if (ch < nch) {
                float o[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = rs * (g[i][j] - s1 - xh[i][j] * s2);
                *reinterpret_cast<uint4*>(dx + (size_t)row * D + ch * 8) = pack8(o);
                if (dx_drop) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const unsigned long long idx = (unsigned long long)row * D + ch * 8 + j;
                        o[j] = dropout_keep(seed, idx, thresh) ? o[j] * dscale : 0.f;
                    }
                    *reinterpret_cast<uint4*>(dx_drop + (size_t)row * D + ch * 8) = pack8(o);
                }
            }
        }
    }
    if (dgamma || partials) {
        // reduce the 4 waves of the block in LDS, then one atomic (or one partial row) per column per block
        __shared__ float red[2][4][LN_MAXCH * 64 * 8];
        const int w = threadIdx.x >> 6;
#pragma unroll
        for (int i = 0; i < LN_MAXCH; ++i) {
            const int ch = lane + 64 * i;
            if (ch < nch) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    red[0][w][ch * 8 + j] = ag[i][j];
                    red[1][w][ch * 8 + j] = ab[i][j];
                }
            }
        }
        __syncthreads();
        for (int c = threadIdx.x; c < D; c += 256) {
            const float sg = (red[0][0][c] + red[0][1][c]) + (red[0][2][c] + red[0][3][c]);
            const float sb = (red[1][0][c] + red[1][1][c]) + (red[1][2][c] + red[1][3][c]);
            if (partials) {               // [2][blocks][D]: folded later by toist_splitk_reduce_batch (no contended atomics)
                partials[(size_t)blockIdx.x * D + c] = sg;
                partials[((size_t)gridDim.x + blockIdx.x) * D + c] = sb;
            } else {
                atomicAdd(dgamma + c, sg);
                atomicAdd(dbeta + c, sb);
            }
        }
    }
}

// ---------------------------------------------------------------------------------- softmax
constexpr int SM_MAXCH = 4;  // Sk <= 2048

// scores/probabilities are [nbatch*H, Sq, ld] bf16 (ld >= roundup8(Sk), pad columns written as 0).
// key_pad [nbatch, Sk] uint8 (1 = padded key -> -inf), may be null.
__global__ __launch_bounds__(256) void softmax_fwd_kernel(const bf16_t* __restrict__ s, const unsigned char* __restrict__ key_pad,
                                                           int rows, int H, int Sq, int Sk, int ld,
                                                           bf16_t* __restrict__ p, bf16_t* __restrict__ p_drop,
                                                           float drop_p, unsigned long long seed,
                                                           const unsigned long long* __restrict__ seed_dev) {
    if (seed_dev) seed += *seed_dev;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int b = row / (H * Sq);
    const int nch = ld >> 3;
    float v[SM_MAXCH][8];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < SM_MAXCH; ++i) {
        const int ch = lane + 64 * i;
        if (ch < nch) {
            unpack8(*reinterpret_cast<const uint4*>(s + (size_t)row * ld + ch * 8), v[i]);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = ch * 8 + j;
                const bool dead = (k >= Sk) || (key_pad && key_pad[(size_t)b * Sk + k]);
                if (dead) v[i][j] = -INFINITY;
                mx = fmaxf(mx, v[i][j]);
            }
        }
    }
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < SM_MAXCH; ++i) {
        const int ch = lane + 64 * i;
        if (ch < nch) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { v[i][j] = __expf(v[i][j] - mx); sum += v[i][j]; }
        }
    }
    sum = wave_sum(sum);
    const float inv = 1.f / sum;
    const unsigned thresh = p_drop ? (unsigned)(drop_p * 4294967296.0) : 0u;
    const float dscale = p_drop ? 1.f / (1.f - drop_p) : 1.f;
#pragma unroll
    for (int i = 0; i < SM_MAXCH; ++i) {
        const int ch = lane + 64 * i;
        if (ch < nch) {
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = v[i][j] * inv;
            *reinterpret_cast<uint4*>(p + (size_t)row * ld + ch * 8) = pack8(o);
            if (p_drop) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const unsigned long long idx = (unsigned long long)row * ld + ch * 8 + j;
                    o[j] = dropout_keep(seed, idx, thresh) ? o[j] * dscale : 0.f;
                }
                *reinterpret_cast<uint4*>(p_drop + (size_t)row * ld + ch * 8) = pack8(o);
            }
        }
    }
}

// ds = p * (m*dp - sum_j p_j*m_j*dp_j), m = dropout keep-mask / (1-p_drop) (1 when no dropout)
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const bf16_t* __restrict__ p, const bf16_t* __restrict__ dp, int rows,
                                                           int Sk, int ld, bf16_t* __restrict__ ds, float drop_p,
                                                           unsigned long long seed, const unsigned long long* __restrict__ seed_dev) {
    if (seed_dev) seed += *seed_dev;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nch = ld >> 3;
    const bool drop = drop_p > 0.f;
    const unsigned thresh = drop ? (unsigned)(drop_p * 4294967296.0) : 0u;
    const float dscale = drop ? 1.f / (1.f - drop_p) : 1.f;
    float pv[SM_MAXCH][8], gv[SM_MAXCH][8];
    float dot = 0.f;
#pragma unroll
    for (int i = 0; i < SM_MAXCH; ++i) {
        const int ch = lane + 64 * i;
        if (ch < nch) {
            unpack8(*reinterpret_cast<const uint4*>(p + (size_t)row * ld + ch * 8), pv[i]);
            unpack8(*reinterpret_cast<const uint4*>(dp + (size_t)row * ld + ch * 8), gv[i]);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = ch * 8 + j;
                if (k >= Sk) { pv[i][j] = 0.f; gv[i][j] = 0.f; }
                if (drop) {
                    const unsigned long long idx = (unsigned long long)row * ld + k;
                    gv[i][j] = dropout_keep(seed, idx, thresh) ? gv[i][j] * dscale : 0.f;
                }
                dot += pv[i][j] * gv[i][j];
            }
        }
    }
    dot = wave_sum(dot);
#pragma unroll
    for (int i = 0; i < SM_MAXCH; ++i) {
        const int ch = lane + 64 * i;
        if (ch < nch) {
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = pv[i][j] * (gv[i][j] - dot);
            *reinterpret_cast<uint4*>(ds + (size_t)row * ld + ch * 8) = pack8(o);
        }
    }
}

// ---------------------------------------------------------------------------------- column sum
// out[n] += sum_m g[m][n]  (bias gradient).  Block = 32 column-chunks (8 columns, 16-byte loads) x 8 row
// lanes over a 256-row slab; LDS reduction over the row lanes, one atomic per column per block.
// column sums of a tall, narrow matrix (N <= 128 columns, e.g. the bias gradient of a 16-channel convolution over 20 M
// pixels): all 256 threads stream 16-byte chunks, thread t always lands on column chunk t % (N/8); one fp32 atomic per
// column per 8192-row block
constexpr int CS_ROWS = 8192;
__global__ __launch_bounds__(256) void colsum_narrow_kernel(const bf16_t* __restrict__ g, long long M, int N, float* __restrict__ out) {
    __shared__ float red[256][9];
    const int cpr = N >> 3;                         // 16-byte chunks per row; divides 256
    const int rpp = 256 / cpr;                      // rows per pass
    const int cx = threadIdx.x % cpr, ry = threadIdx.x / cpr;
    const long long m_beg = (long long)blockIdx.x * CS_ROWS;
    const long long m_end = (m_beg + CS_ROWS < M) ? m_beg + CS_ROWS : M;
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (long long m = m_beg + ry; m < m_end; m += rpp) {
        float v[8];
        unpack8(*reinterpret_cast<const uint4*>(g + m * N + cx * 8), v);
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] += v[j];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) red[threadIdx.x][j] = a[j];
    __syncthreads();
    if ((int)threadIdx.x < N) {
        const int c = threadIdx.x, cc = c >> 3, j = c & 7;
        float t = 0.f;
        for (int r = 0; r < rpp; ++r) t += red[r * cpr + cc][j];
        atomicAdd(out + c, t);
    }
}

__global__ __launch_bounds__(256) void colsum_kernel(const bf16_t* __restrict__ g, int M, int N, int ld, float* __restrict__ out) {
    __shared__ float red[8][256];
    const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
    const int n = blockIdx.x * 256 + cx * 8;
    const int m_beg = blockIdx.y * 256;
    const int m_end = (m_beg + 256 < M) ? m_beg + 256 : M;
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (n < N) {
        const bool vec = (n + 8 <= N) && ((ld & 7) == 0) && ((((size_t)g) & 15) == 0);
        for (int m = m_beg + ry; m < m_end; m += 8) {
            if (vec) {
                float v[8];
                unpack8(*reinterpret_cast<const uint4*>(g + (size_t)m * ld + n), v);
#pragma unroll
                for (int j = 0; j < 8; ++j) a[j] += v[j];
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) if (n + j < N) a[j] += bf2f(g[(size_t)m * ld + n + j]);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) red[ry][cx * 8 + j] = a[j];
    __syncthreads();
    const int c = threadIdx.x;
    if (blockIdx.x * 256 + c < N) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) t += red[r][c];
        atomicAdd(out + blockIdx.x * 256 + c, t);
    }
}

// ---------------------------------------------------------------------------------- elementwise
// out = a + b ; b is broadcast with period `bperiod` elements (bperiod == n -> plain add)
__global__ __launch_bounds__(256) void add_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b, long long n8,
                                                   long long bperiod8, bf16_t* __restrict__ out) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
        float x[8], y[8];
        unpack8(reinterpret_cast<const uint4*>(a)[i], x);
        unpack8(reinterpret_cast<const uint4*>(b)[i % bperiod8], y);
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] += y[j];
        reinterpret_cast<uint4*>(out)[i] = pack8(x);
    }
}

// out = dropout(x) with the (seed, index) mask used by the GEMM epilogue / LN backward
__global__ __launch_bounds__(256) void dropout_kernel(const bf16_t* __restrict__ x, long long n8, float p, unsigned long long seed,
                                                       const unsigned long long* __restrict__ seed_dev, bf16_t* __restrict__ out) {
    if (seed_dev) seed += *seed_dev;
    const unsigned thresh = (unsigned)(p * 4294967296.0);
    const float sc = 1.f / (1.f - p);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
        float v[8];
        unpack8(reinterpret_cast<const uint4*>(x)[i], v);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = dropout_keep(seed, (unsigned long long)i * 8 + j, thresh) ? v[j] * sc : 0.f;
        reinterpret_cast<uint4*>(out)[i] = pack8(v);
    }
}


// ---------------------------------------------------------------------------------- embeddings
// RoBERTa input embedding (HF RobertaEmbeddings, reached from transformer.py:130):
// out[t] = word[ids[t]] + type[0] + pos[pos_ids[t]]  (fp32 tables -> bf16 activation).
__global__ __launch_bounds__(256) void embed_fwd_kernel(const long long* __restrict__ ids, const long long* __restrict__ pos_ids,
                                                         const float* __restrict__ word, const float* __restrict__ pos,
                                                         const float* __restrict__ type0, int n, int D, bf16_t* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= n) return;
    const float* w = word + ids[t] * (long long)D;
    const float* p = pos + pos_ids[t] * (long long)D;
    for (int c = lane * 2; c < D; c += 128) {
        const float a = (w[c] + type0[c]) + p[c];
        const float b = (w[c + 1] + type0[c + 1]) + p[c + 1];
        *reinterpret_cast<unsigned*>(out + (size_t)t * D + c) = pack2bf(a, b);
    }
}
// scatter-add of the embedding gradient into the (dense, fp32) table gradients
__global__ __launch_bounds__(256) void embed_bwd_kernel(const bf16_t* __restrict__ g, const long long* __restrict__ ids,
                                                         const long long* __restrict__ pos_ids, int n, int D, long long pad_id,
                                                         float* __restrict__ dword, float* __restrict__ dpos, float* __restrict__ dtype0) {
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= n) return;
    // nn.Embedding(padding_idx=pad_id) never accumulates a gradient into the padding row
    float* dw = (dword && ids[t] != pad_id) ? dword + ids[t] * (long long)D : nullptr;
    float* dp = (dpos && pos_ids[t] != pad_id) ? dpos + pos_ids[t] * (long long)D : nullptr;
    for (int c = lane; c < D; c += 64) {
        const float v = bf2f(g[(size_t)t * D + c]);
        if (dw) atomicAdd(dw + c, v);
        if (dp) atomicAdd(dp + c, v);
        if (dtype0) atomicAdd(dtype0 + c, v);
    }
}

static inline int grid_for(long long n, int cap = 2048) {
    long long g = (n + 255) / 256;
    if (g < 1) g = 1;
    return (int)(g > cap ? cap : g);
}

}  // namespace toist

using namespace toist;

extern "C" int toist_layernorm_fwd(const void* x, const float* gamma, const float* beta, float eps, int rows, int D,
                                   void* y, float* mean, float* rstd, const void* add, void* y2, void* stream) {
    TOIST_REQUIRE(rows > 0 && D > 0 && (D % 8) == 0 && D <= 1024, "toist_layernorm_fwd: rows=%d D=%d (D%%8==0, D<=1024)", rows, D);
    TOIST_REQUIRE((add == nullptr) == (y2 == nullptr), "toist_layernorm_fwd: add and y2 come together");
    hipLaunchKernelGGL(layernorm_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, gamma,
                       beta, eps, rows, D, (bf16_t*)y, mean, rstd, (const bf16_t*)add, (bf16_t*)y2);
    return check_launch("toist_layernorm_fwd");
}

// grid of the partial-sum form: 8 rows per block up to 512 blocks (measured: 3328 x 256 in 4.8 us, 13312 x 256 in 9.1 us)
extern "C" int toist_layernorm_bwd_blocks(int rows) {
    int blocks = (rows + 7) / 8;
    if (blocks > 512) blocks = 512;
    return blocks < 1 ? 1 : blocks;
}

extern "C" int toist_layernorm_bwd(const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma,
                                   int rows, int D, void* dx, float* dgamma, float* dbeta, void* dx_drop, float drop_p,
                                   uint64_t seed, const uint64_t* seed_dev, float* partials, int partial_blocks, void* stream) {
    TOIST_REQUIRE(rows > 0 && D > 0 && (D % 8) == 0 && D <= 1024, "toist_layernorm_bwd: rows=%d D=%d", rows, D);
    TOIST_REQUIRE((dgamma == nullptr) == (dbeta == nullptr), "toist_layernorm_bwd: dgamma/dbeta must both be set or null");
    TOIST_REQUIRE(partials == nullptr || (dgamma == nullptr && partial_blocks > 0), "toist_layernorm_bwd: partials replace dgamma / dbeta and need a block count");
    // With atomics: few blocks (>= 4 rows per wave), the parameter-gradient atomics are what the kernel waits for (3328 x 256: 10.1 us
    // with 128 blocks, 24 us with 1024; 5.7 us without them).  With `partials` the caller picks the grid (toist_layernorm_bwd_blocks).
    int blocks = (rows + 15) / 16;
    if (blocks > 128) blocks = 128;
    if (blocks < 1) blocks = 1;
    if (partials) blocks = partial_blocks;
    hipLaunchKernelGGL(layernorm_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy,
                       (const bf16_t*)x, mean, rstd, gamma, rows, D, (bf16_t*)dx, dgamma, dbeta, (bf16_t*)dx_drop, drop_p,
                       (unsigned long long)seed, (const unsigned long long*)seed_dev, partials);
    return check_launch("toist_layernorm_bwd");
}

extern "C" int toist_softmax_fwd(const void* scores, const uint8_t* key_pad, int nbatch, int H, int Sq, int Sk, int ld,
                                 void* p, void* p_drop, float drop_p, uint64_t seed, const uint64_t* seed_dev, void* stream) {
    TOIST_REQUIRE(nbatch > 0 && H > 0 && Sq > 0 && Sk > 0, "toist_softmax_fwd: bad shape");
    TOIST_REQUIRE((ld % 8) == 0 && ld >= Sk && ld <= 2048, "toist_softmax_fwd: ld=%d Sk=%d (ld%%8==0, Sk<=ld<=2048)", ld, Sk);
    const int rows = nbatch * H * Sq;
    hipLaunchKernelGGL(softmax_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)scores, key_pad,
                       rows, H, Sq, Sk, ld, (bf16_t*)p, (bf16_t*)p_drop, drop_p, (unsigned long long)seed, (const unsigned long long*)seed_dev);
    return check_launch("toist_softmax_fwd");
}

extern "C" int toist_softmax_bwd(const void* p, const void* dp, int rows, int Sk, int ld, void* ds, float drop_p, uint64_t seed,
                                 const uint64_t* seed_dev, void* stream) {
    TOIST_REQUIRE(rows > 0 && Sk > 0 && (ld % 8) == 0 && ld >= Sk && ld <= 2048, "toist_softmax_bwd: bad shape");
    hipLaunchKernelGGL(softmax_bwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)p,
                       (const bf16_t*)dp, rows, Sk, ld, (bf16_t*)ds, drop_p, (unsigned long long)seed, (const unsigned long long*)seed_dev);
    return check_launch("toist_softmax_bwd");
}

extern "C" int toist_colsum(const void* g, int M, int N, int ld, float* out, void* stream) {
    TOIST_REQUIRE(M > 0 && N > 0 && ld >= N, "toist_colsum: bad shape");
    if (ld == N && N <= 128 && (N & 7) == 0 && (256 % (N >> 3)) == 0 && M >= 4 * CS_ROWS && ((((size_t)g) & 15) == 0)) {
        hipLaunchKernelGGL(colsum_narrow_kernel, dim3((unsigned)((M + CS_ROWS - 1) / CS_ROWS)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)g,
                           (long long)M, N, out);
        return check_launch("toist_colsum(narrow)");
    }
    hipLaunchKernelGGL(colsum_kernel, dim3((N + 255) / 256, (M + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)g, M,
                       N, ld, out);
    return check_launch("toist_colsum");
}

extern "C" int toist_add_bf16(const void* a, const void* b, int64_t n, int64_t b_period, void* out, void* stream) {
    TOIST_REQUIRE(n > 0 && (n % 8) == 0 && b_period > 0 && (b_period % 8) == 0 && (n % b_period) == 0,
                  "toist_add_bf16: n and b_period must be multiples of 8 and n %% b_period == 0");
    hipLaunchKernelGGL(add_kernel, dim3(grid_for(n / 8)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)a, (const bf16_t*)b,
                       (long long)(n / 8), (long long)(b_period / 8), (bf16_t*)out);
    return check_launch("toist_add_bf16");
}

extern "C" int toist_dropout_bf16(const void* x, int64_t n, float p, uint64_t seed, const uint64_t* seed_dev, void* out, void* stream) {
    TOIST_REQUIRE(n > 0 && (n % 8) == 0 && p >= 0.f && p < 1.f, "toist_dropout_bf16: bad args");
    hipLaunchKernelGGL(dropout_kernel, dim3(grid_for(n / 8)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (long long)(n / 8),
                       p, (unsigned long long)seed, (const unsigned long long*)seed_dev, (bf16_t*)out);
    return check_launch("toist_dropout_bf16");
}

extern "C" int toist_embed_fwd(const int64_t* ids, const int64_t* pos_ids, const float* word, const float* pos, const float* type0,
                               int n, int D, void* out, void* stream) {
    TOIST_REQUIRE(n > 0 && D > 0 && (D % 2) == 0, "toist_embed_fwd: bad shape");
    hipLaunchKernelGGL(embed_fwd_kernel, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const long long*)ids,
                       (const long long*)pos_ids, word, pos, type0, n, D, (bf16_t*)out);
    return check_launch("toist_embed_fwd");
}

extern "C" int toist_embed_bwd(const void* g, const int64_t* ids, const int64_t* pos_ids, int n, int D, int64_t pad_id, float* dword,
                               float* dpos, float* dtype0, void* stream) {
    TOIST_REQUIRE(n > 0 && D > 0, "toist_embed_bwd: bad shape");
    hipLaunchKernelGGL(embed_bwd_kernel, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)g, (const long long*)ids,
                       (const long long*)pos_ids, n, D, (long long)pad_id, dword, dpos, dtype0);
    return check_launch("toist_embed_bwd");
}
