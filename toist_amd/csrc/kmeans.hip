// Device-resident k-means for the noun-pronoun distillation step (BASELINE configs[4]).
//
// Replaces models/kmeans.py:21-96 as ClusterCriterion.memory_cluster calls it (/root/reference/models/mdetr.py:213-234): per
// sample, Lloyd iterations over the task's memory bank ([1024, 256] fp32) from the task's current centres (K = 3) until
// (sum_k |c_k' - c_k|)^2 < tol, the centres are stored back, and the centre nearest to the sample's own feature is picked.  The
// reference (and round 1 here) reads the shift back to the host after EVERY iteration; here the whole loop runs inside one
// workgroup -- one workgroup per distinct task of the batch, walking that task's samples in batch order so that a later sample
// starts from the centres the earlier one left, exactly like the sequential reference -- and the host never looks.
// Work per iteration: two passes over the 1 MB bank (L2-resident) = ~20 us on one CU; neither roofline applies to 1-8 workgroups.
#include "common.h"

namespace toist {

constexpr int KM_THREADS = 1024, KM_MAXK = 8, KM_MAXD = 256;

__global__ __launch_bounds__(KM_THREADS) void kmeans_kernel(const float* __restrict__ banks, long long bank_stride, float* __restrict__ centers_all,
                                                            long long centers_stride, const int* __restrict__ group_task,
                                                            const int* __restrict__ group_off, const int* __restrict__ members,
                                                            const float* __restrict__ features, int N, int D, int K, float tol, int max_iter,
                                                            int* __restrict__ pick_out, float* __restrict__ center_out, int* __restrict__ iters_out) {
    __shared__ float c[KM_MAXK][KM_MAXD];            // current centres
    __shared__ float part[4][KM_MAXK][KM_MAXD];      // per-quarter member sums
    __shared__ int cnt_part[4][KM_MAXK];
    __shared__ float red[KM_THREADS / 64];
    __shared__ float shift_sh;
    extern __shared__ unsigned char choice[];        // [N]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = blockIdx.x, task = group_task[g];
    const float* X = banks + (long long)task * bank_stride;
    float* cg = centers_all + (long long)task * centers_stride;
    for (int i = tid; i < K * D; i += KM_THREADS) c[i / D][i % D] = cg[i];
    __syncthreads();
    const int d_of = tid & 255, quarter = tid >> 8;  // update pass: thread = (feature, quarter of the points); needs D <= 256
    for (int mi = group_off[g]; mi < group_off[g + 1]; ++mi) {
        const int sample = members[mi];
        int it = 0;
        for (; it < max_iter; ++it) {
            // ---- assignment: one wavefront per point, lanes across the features ----
            for (int n = wave; n < N; n += KM_THREADS / 64) {
                float dist[KM_MAXK];
#pragma unroll
                for (int kk = 0; kk < KM_MAXK; ++kk) dist[kk] = 0.f;
                for (int d = lane; d < D; d += 64) {
                    const float x = X[(long long)n * D + d];
#pragma unroll
                    for (int kk = 0; kk < KM_MAXK; ++kk)
                        if (kk < K) { const float t = x - c[kk][d]; dist[kk] += t * t; }
                }
                int best = 0;
                float bd = 0.f;
#pragma unroll
                for (int kk = 0; kk < KM_MAXK; ++kk) {
                    if (kk < K) {
                        const float v = wave_sum(dist[kk]);
                        if (kk == 0 || v < bd) { bd = v; best = kk; }      // first minimum wins, like torch.argmin
                    }
                }
                if (lane == 0) choice[n] = (unsigned char)best;
            }
            __syncthreads();
            // ---- update: member sums per (cluster, feature), four quarters of the points in parallel ----
            {
                float s[KM_MAXK];
                int cn[KM_MAXK];
#pragma unroll
                for (int kk = 0; kk < KM_MAXK; ++kk) { s[kk] = 0.f; cn[kk] = 0; }
                const int per = (N + 3) / 4, n0 = quarter * per, n1 = min(N, n0 + per);
                if (d_of < D) {
                    for (int n = n0; n < n1; ++n) {
                        const int ch = choice[n];
                        const float x = X[(long long)n * D + d_of];
#pragma unroll
                        for (int kk = 0; kk < KM_MAXK; ++kk)
                            if (kk == ch) { s[kk] += x; ++cn[kk]; }
                    }
#pragma unroll
                    for (int kk = 0; kk < KM_MAXK; ++kk)
                        if (kk < K) { part[quarter][kk][d_of] = s[kk]; if (d_of == 0) cnt_part[quarter][kk] = cn[kk]; }
                }
            }
            __syncthreads();
            // threads 0 .. K*D-1 finish one (cluster, feature) each; the squared moves go back to LDS for the per-cluster norms
            float nw = 0.f, df = 0.f;
            int fk = 0, fd = 0;
            if (tid < K * D) {
                fk = tid / D; fd = tid - fk * D;
                const int count = cnt_part[0][fk] + cnt_part[1][fk] + cnt_part[2][fk] + cnt_part[3][fk];
                const float old = c[fk][fd];
                nw = old;                                             // an empty cluster keeps its centre (kmeans.py:72)
                if (count > 0) nw = (((part[0][fk][fd] + part[1][fk][fd]) + part[2][fk][fd]) + part[3][fk][fd]) / (float)count;
                df = (nw - old) * (nw - old);
            }
            __syncthreads();                                          // every read of part[] is done
            if (tid < K * D) { c[fk][fd] = nw; part[0][fk][fd] = df; }
            __syncthreads();
            if (tid < K) {                                            // |c_k' - c_k| per cluster (D <= 256 terms)
                float t = 0.f;
                for (int d = 0; d < D; ++d) t += part[0][tid][d];
                red[tid] = sqrtf(t);
            }
            __syncthreads();
            if (tid == 0) {
                float sh = 0.f;
                for (int kk = 0; kk < K; ++kk) sh += red[kk];
                shift_sh = sh;
            }
            __syncthreads();
            if (shift_sh * shift_sh < tol) { ++it; break; }
        }
        // ---- centres back to the task's slot; pick the centre nearest to this sample's feature ----
        for (int i = tid; i < K * D; i += KM_THREADS) cg[i] = c[i / D][i % D];
        if (wave == 0) {
            const float* f = features + (long long)sample * D;
            int best = 0;
            float bd = 0.f;
            for (int kk = 0; kk < K; ++kk) {
                float t = 0.f;
                for (int d = lane; d < D; d += 64) { const float u = f[d] - c[kk][d]; t += u * u; }
                t = wave_sum(t);
                if (kk == 0 || t < bd) { bd = t; best = kk; }
            }
            if (lane == 0) { pick_out[sample] = best; if (iters_out) iters_out[sample] = it; }
            for (int d = lane; d < D; d += 64) center_out[(long long)sample * D + d] = c[best][d];
        }
        __syncthreads();
    }
}

// dst[dst_row[i]] = src[src_row[i]] for the rows i with dst_row[i] >= 0 and src_row[i] >= 0 (rows of `d` floats; distinct live destinations).
// The memory-bank replacement of the distillation step under hipGraph replay: which slots are live is decided on the device (an LSAP pair table of
// fixed capacity, its status word), so the write must skip the dead ones instead of sending them to a scratch row the bank does not have.
__global__ __launch_bounds__(256) void scatter_rows_kernel(const float* __restrict__ src, const long long* __restrict__ src_row, float* __restrict__ dst,
                                                            const long long* __restrict__ dst_row, int m, int d) {
    const int i = blockIdx.x;
    const long long sr = src_row[i], dr = dst_row[i];
    if (i >= m || sr < 0 || dr < 0) return;
    for (int e = threadIdx.x; e < d; e += 256) dst[dr * d + e] = src[sr * d + e];
}

}  // namespace toist

using namespace toist;

extern "C" int toist_kmeans(const float* banks, int64_t bank_stride, float* centers, int64_t centers_stride, const int32_t* group_task,
                            const int32_t* group_off, const int32_t* members, int n_groups, const float* features, int N, int D, int K, float tol,
                            int max_iter, int32_t* pick, float* chosen_center, int32_t* iters, void* stream) {
    TOIST_REQUIRE(banks && centers && group_task && group_off && members && features && pick && chosen_center && n_groups > 0, "toist_kmeans: bad args");
    TOIST_REQUIRE(N > 0 && N <= 32768 && D > 0 && D <= KM_MAXD && D > 0 && K > 0 && K <= KM_MAXK && K * D <= KM_THREADS && max_iter > 0 && tol >= 0.f,
                  "toist_kmeans: N <= 32768, D <= %d, K <= %d, K * D <= %d (N=%d D=%d K=%d)", KM_MAXD, KM_MAXK, KM_THREADS, N, D, K);
    hipLaunchKernelGGL(kmeans_kernel, dim3(n_groups), dim3(KM_THREADS), (size_t)((N + 15) & ~15), (hipStream_t)stream, banks, (long long)bank_stride, centers,
                       (long long)centers_stride, group_task, group_off, members, features, N, D, K, tol, max_iter, pick, chosen_center, iters);
    return check_launch("toist_kmeans");
}

extern "C" int toist_scatter_rows_f32(const float* src, const int64_t* src_row, float* dst, const int64_t* dst_row, int m, int d, void* stream) {
    using namespace toist;
    TOIST_REQUIRE(src && src_row && dst && dst_row && m > 0 && d > 0, "toist_scatter_rows_f32: bad arguments");
    hipLaunchKernelGGL(scatter_rows_kernel, dim3(m), dim3(256), 0, (hipStream_t)stream, src, (const long long*)src_row, dst, (const long long*)dst_row, m, d);
    return check_launch("toist_scatter_rows_f32");
}
