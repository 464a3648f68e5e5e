// How fast does MI355X absorb a bf16 [M][N] matrix written as TILES (one workgroup per BM x BN tile, a wave instruction = 8 rows
// x 128 B when BN = 64) compared with a streaming write?  (The epilogue of the 1x1-convolution GEMMs writes 64 x 64 tiles.)
//   hipcc --offload-arch=gfx950 -O3 tools/probe/store_pattern.hip -o build/store_pattern && build/store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int BM, int BN, bool NT, bool READ>
__global__ __launch_bounds__(256) void tile_store(uint4* __restrict__ out, const uint4* __restrict__ in, int M, int N, int order) {
    constexpr int CPR = BN / 8;                 // 16-byte chunks per tile row
    constexpr int ROWS = 256 / CPR;             // rows per pass
    const int nt_n = N / BN, nt_m = M / BM;
    int t = blockIdx.x;
    int tm, tn;
    if (order == 0) { tm = t / nt_n; tn = t - tm * nt_n; }          // N fastest: neighbours in N run together
    else { tn = t / nt_m; tm = t - tn * nt_m; }                      // M fastest
    const int c = threadIdx.x % CPR, r0 = threadIdx.x / CPR;
    for (int r = r0; r < BM; r += ROWS) {
        const size_t idx = ((size_t)(tm * BM + r) * N + tn * BN) / 8 + c;
        uint4 v = make_uint4(r, c, t, 7);
        if (READ) { const uint4 u = in[idx]; v.x += u.x; v.y ^= u.y; }
        typedef __attribute__((ext_vector_type(4))) unsigned int u4;
        if (NT) __builtin_nontemporal_store(u4{v.x, v.y, v.z, v.w}, reinterpret_cast<u4*>(out + idx)); else out[idx] = v;
    }
}

template <typename F> static float time_us(F f, int iters = 20) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a); for (int i = 0; i < iters; ++i) f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms * 1000.f / iters;
}

int main() {
    const int shapes[][2] = {{12800, 1024}, {51200, 512}, {204800, 256}};
    for (auto& sh : shapes) {
        const int M = sh[0], N = sh[1];
        const size_t bytes = (size_t)M * N * 2;
        // rotate over 12 buffers (> the 256 MB Infinity Cache for the larger shapes) like the model's distinct activations
        std::vector<uint4*> outs(12), ins(12);
        for (int i = 0; i < 12; ++i) { hipMalloc(&outs[i], bytes); hipMalloc(&ins[i], bytes); hipMemset(ins[i], 1, bytes); }
        int rot = 0;
        auto run = [&](auto kern, int bm, int bn, int order) {
            return time_us([&] { rot = (rot + 1) % 12; hipLaunchKernelGGL(kern, dim3((M / bm) * (N / bn)), dim3(256), 0, 0, outs[rot], ins[rot], M, N, order); });
        };
        printf("%d x %d (%.0f MB)\n", M, N, bytes / 1e6);
#define ROW(BM, BN) \
        printf("  tile %3dx%-4d  write: Nfast %6.1f  Mfast %6.1f  nt %6.1f | read+write: Nfast %6.1f  Mfast %6.1f  nt %6.1f us\n", BM, BN, \
               run(tile_store<BM, BN, false, false>, BM, BN, 0), run(tile_store<BM, BN, false, false>, BM, BN, 1), run(tile_store<BM, BN, true, false>, BM, BN, 0), \
               run(tile_store<BM, BN, false, true>, BM, BN, 0), run(tile_store<BM, BN, false, true>, BM, BN, 1), run(tile_store<BM, BN, true, true>, BM, BN, 0));
        ROW(64, 64) ROW(64, 128) ROW(64, 256) ROW(32, 256) ROW(16, 256) ROW(128, 64) ROW(32, 64)
        if (N >= 512) { ROW(16, 512) ROW(8, 512) }
        if (N >= 1024) { ROW(8, 1024) ROW(4, 1024) }
        for (int i = 0; i < 12; ++i) { hipFree(outs[i]); hipFree(ins[i]); }
    }
    return 0;
}
