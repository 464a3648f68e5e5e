// Optimizer tail of the TOIST training step as ONE multi-tensor pass (reference: engine.py:87-101 --
// clip_grad_norm_(max_norm) -> AdamW.step() -> update_ema(); util/optim.py:9-26; main.py:351-392 parameter groups):
//   sqnorm_kernel      sum of squares of every gradient chunk                (reads g once)
//   finish_norm_kernel total norm, clip coefficient, step count, bias corrections (one block, device-side state)
//   adamw_ema_kernel   g*clip -> decoupled weight decay -> Adam moments -> parameter -> EMA -> bf16 compute copy
//                      (reads p,g,m,v,ema; writes p,m,v,ema,w_bf16: 38 B/parameter instead of three library sweeps
//                      plus one cast kernel per weight tensor)
// HBM-bound; every access is a 16-byte vector.  Tensors are described by a device-resident table, work is cut
// into fixed chunks so one launch covers all ~800 tensors (block b -> chunks[b] = {tensor, chunk index}).
#include "common.h"

namespace toist {

constexpr int OPT_CHUNK = 8192;  // elements per block (256 threads x 8 float4)

__device__ __forceinline__ float block_sum_256(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) red[wave] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void sqnorm_kernel(const toist_opt_tensor* __restrict__ T, const int64_t* __restrict__ G,
                                                     const int2* __restrict__ chunks, float* __restrict__ partial) {
    __shared__ float red[4];
    const int2 ch = chunks[blockIdx.x];
    const float* g = reinterpret_cast<const float*>(G[ch.x]);
    float acc = 0.f;
    if (g != nullptr) {
        const long long beg = (long long)ch.y * OPT_CHUNK;
        long long end = beg + OPT_CHUNK;
        if (end > T[ch.x].numel) end = T[ch.x].numel;
        const float* gp = g + beg;
        const int n = (int)(end - beg);
        if (((size_t)gp & 15) == 0) {
            const int n4 = n >> 2;
            for (int i = threadIdx.x; i < n4; i += 256) {
                const float4 x = reinterpret_cast<const float4*>(gp)[i];
                acc += x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w;
            }
            for (int i = (n4 << 2) + threadIdx.x; i < n; i += 256) acc += gp[i] * gp[i];
        } else {
            for (int i = threadIdx.x; i < n; i += 256) acc += gp[i] * gp[i];
        }
    }
    const float s = block_sum_256(acc, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// (one block on the step's critical path between the gradient norm and the update: 1024 threads, four independent loads in flight per thread --
// 22 600 partial sums took 30 us with 256 threads walking them one load at a time)
__global__ __launch_bounds__(1024) void finish_norm_kernel(const float* __restrict__ partial, int n, float max_norm, float beta1,
                                                           float beta2, toist_opt_state* __restrict__ st) {
    __shared__ double red[1024];
    double acc = 0.0;
    int i = threadIdx.x;
    for (; i + 3 * 1024 < n; i += 4 * 1024) {
        const float a = partial[i], b = partial[i + 1024], c = partial[i + 2 * 1024], d = partial[i + 3 * 1024];
        acc += ((double)a + (double)b) + ((double)c + (double)d);
    }
    for (; i < n; i += 1024) acc += (double)partial[i];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float norm = (float)sqrt(red[0]);
        float coef = 1.f;
        if (max_norm > 0.f) {  // torch.nn.utils.clip_grad_norm_: coef = clamp(max_norm / (norm + 1e-6), max=1)
            coef = max_norm / (norm + 1e-6f);
            if (coef > 1.f) coef = 1.f;  // NaN stays NaN, as in torch
        }
        const int step = st->step + 1;
        st->step = step;
        st->grad_norm = norm;
        st->clip_coef = coef;
        st->bias1 = 1.f - powf(beta1, (float)step);
        st->bias2_sqrt = sqrtf(1.f - powf(beta2, (float)step));
    }
}

__device__ __forceinline__ void adamw_one(float& p, float g, float& m, float& v, float lr, float wd, float b1, float b2, float eps,
                                          float step_size, float bias2_sqrt) {
    p -= lr * wd * p;                       // decoupled weight decay
    m += (g - m) * (1.f - b1);              // exp_avg.lerp_(grad, 1 - beta1)
    v = b2 * v + (1.f - b2) * g * g;
    const float denom = sqrtf(v) / bias2_sqrt + eps;
    p -= step_size * (m / denom);
}

__global__ __launch_bounds__(256) void adamw_ema_kernel(const toist_opt_tensor* __restrict__ T, const int64_t* __restrict__ G,
                                                        const int2* __restrict__ chunks, const toist_opt_group* __restrict__ groups,
                                                        const toist_opt_state* __restrict__ st, float b1, float b2, float eps,
                                                        float decay, int n_chunks) {
  // gridDim.x < n_chunks: a slim launch (the late groups' update beside the next forward pass) walks the chunks with few workgroups
  for (int cb = blockIdx.x; cb < n_chunks; cb += gridDim.x) {
    const int2 ch = chunks[cb];
    const toist_opt_tensor t = T[ch.x];
    const float* g = reinterpret_cast<const float*>(G[ch.x]);
    const long long beg = (long long)ch.y * OPT_CHUNK;
    long long end = beg + OPT_CHUNK;
    if (end > t.numel) end = t.numel;
    const int n = (int)(end - beg);
    const float coef = st->clip_coef, bias2_sqrt = st->bias2_sqrt;
    const toist_opt_group gr = groups[t.group];
    const float lr = gr.lr, wd = gr.weight_decay, step_size = lr / st->bias1;
    float* p = t.p + beg;
    const bool upd = g != nullptr && t.m != nullptr;
    const bool vec = (((size_t)p | (size_t)(g ? g + beg : nullptr) | (size_t)(t.m ? t.m + beg : nullptr) | (size_t)(t.v ? t.v + beg : nullptr) |
                       (size_t)(t.ema ? t.ema + beg : nullptr)) & 15) == 0 &&
                     (t.w == nullptr || (((size_t)(t.w + beg)) & 7) == 0);
    const int n4 = vec ? (n >> 2) : 0;
    for (int i = threadIdx.x; i < n4; i += 256) {
        float4 pv = reinterpret_cast<float4*>(p)[i];
        if (upd) {
            float4 gv = reinterpret_cast<const float4*>(g + beg)[i];
            float4 mv = reinterpret_cast<float4*>(t.m + beg)[i];
            float4 vv = reinterpret_cast<float4*>(t.v + beg)[i];
            adamw_one(pv.x, gv.x * coef, mv.x, vv.x, lr, wd, b1, b2, eps, step_size, bias2_sqrt);
            adamw_one(pv.y, gv.y * coef, mv.y, vv.y, lr, wd, b1, b2, eps, step_size, bias2_sqrt);
            adamw_one(pv.z, gv.z * coef, mv.z, vv.z, lr, wd, b1, b2, eps, step_size, bias2_sqrt);
            adamw_one(pv.w, gv.w * coef, mv.w, vv.w, lr, wd, b1, b2, eps, step_size, bias2_sqrt);
            reinterpret_cast<float4*>(p)[i] = pv;
            reinterpret_cast<float4*>(t.m + beg)[i] = mv;
            reinterpret_cast<float4*>(t.v + beg)[i] = vv;
        }
        if (t.ema != nullptr) {
            float4 ev = reinterpret_cast<float4*>(t.ema + beg)[i];
            ev.x = ev.x * decay + (1.f - decay) * pv.x;
            ev.y = ev.y * decay + (1.f - decay) * pv.y;
            ev.z = ev.z * decay + (1.f - decay) * pv.z;
            ev.w = ev.w * decay + (1.f - decay) * pv.w;
            reinterpret_cast<float4*>(t.ema + beg)[i] = ev;
        }
        if (upd && t.w != nullptr) {
            float s = 1.f;
            const long long e0 = beg + ((long long)i << 2);
            if (t.row_scale != nullptr) {
                const long long r0 = e0 / t.row_len;
                if ((e0 + 3) / t.row_len == r0) s = t.row_scale[r0];
                else {  // the 4 elements straddle a row boundary (row_len % 4 != 0): scale one by one
                    pv.x *= t.row_scale[e0 / t.row_len];
                    pv.y *= t.row_scale[(e0 + 1) / t.row_len];
                    pv.z *= t.row_scale[(e0 + 2) / t.row_len];
                    pv.w *= t.row_scale[(e0 + 3) / t.row_len];
                }
            }
            reinterpret_cast<uint2*>(t.w + beg)[i] = make_uint2(pack2bf(pv.x * s, pv.y * s), pack2bf(pv.z * s, pv.w * s));
        }
    }
    for (int i = (n4 << 2) + threadIdx.x; i < n; i += 256) {
        float pv = p[i];
        if (upd) {
            float mv = t.m[beg + i], vv = t.v[beg + i];
            adamw_one(pv, g[beg + i] * coef, mv, vv, lr, wd, b1, b2, eps, step_size, bias2_sqrt);
            p[i] = pv;
            t.m[beg + i] = mv;
            t.v[beg + i] = vv;
        }
        if (t.ema != nullptr) t.ema[beg + i] = t.ema[beg + i] * decay + (1.f - decay) * pv;
        if (upd && t.w != nullptr) {
            const float s = t.row_scale ? t.row_scale[(beg + i) / t.row_len] : 1.f;
            t.w[beg + i] = f2bf(pv * s);
        }
    }
  }
}

}  // namespace toist

using namespace toist;

extern "C" int toist_opt_chunk_elems(void) { return OPT_CHUNK; }

extern "C" int toist_opt_sqnorm(const toist_opt_tensor* table, const int64_t* grads, const int32_t* chunks, int n_chunks, float* partial,
                                void* stream) {
    TOIST_REQUIRE(table && grads && chunks && partial && n_chunks > 0, "toist_opt_sqnorm: bad args");
    hipLaunchKernelGGL(sqnorm_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, table, grads, (const int2*)chunks, partial);
    return check_launch("toist_opt_sqnorm");
}

extern "C" int toist_opt_finish_norm(const float* partial, int n_chunks, float max_norm, float beta1, float beta2, toist_opt_state* state,
                                     void* stream) {
    TOIST_REQUIRE(partial && state && n_chunks > 0, "toist_opt_finish_norm: bad args");
    TOIST_REQUIRE(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f, "toist_opt_finish_norm: betas must be in [0, 1)");
    hipLaunchKernelGGL(finish_norm_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, partial, n_chunks, max_norm, beta1, beta2, state);
    return check_launch("toist_opt_finish_norm");
}

extern "C" int toist_opt_adamw_ema_blocks(const toist_opt_tensor* table, const int64_t* grads, const int32_t* chunks, int n_chunks,
                                          const toist_opt_group* groups, const toist_opt_state* state, float beta1, float beta2, float eps,
                                          float ema_decay, int max_blocks, void* stream) {
    TOIST_REQUIRE(table && grads && chunks && groups && state && n_chunks > 0, "toist_opt_adamw_ema: bad args");
    TOIST_REQUIRE(eps > 0.f && ema_decay >= 0.f && ema_decay <= 1.f, "toist_opt_adamw_ema: bad eps / ema_decay");
    const int blocks = (max_blocks > 0 && max_blocks < n_chunks) ? max_blocks : n_chunks;
    hipLaunchKernelGGL(adamw_ema_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, table, grads, (const int2*)chunks, groups, state,
                       beta1, beta2, eps, ema_decay, n_chunks);
    return check_launch("toist_opt_adamw_ema");
}

extern "C" int toist_opt_adamw_ema(const toist_opt_tensor* table, const int64_t* grads, const int32_t* chunks, int n_chunks,
                                   const toist_opt_group* groups, const toist_opt_state* state, float beta1, float beta2, float eps,
                                   float ema_decay, void* stream) {
    return toist_opt_adamw_ema_blocks(table, grads, chunks, n_chunks, groups, state, beta1, beta2, eps, ema_decay, 0, stream);
}
