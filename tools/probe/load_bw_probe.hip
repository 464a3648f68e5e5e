// Probe: how many bytes per clock a CU can pull through (a) LDS-DMA (`buffer_load_dwordx4 ... lds`, bytes in flight bounded by
// the LDS landing zone) and (b) plain vector loads into registers (bytes in flight bounded by VGPRs), at GEMM-like occupancy.
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 load_bw_probe.hip -o /tmp/bw && /tmp/bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(4))) int i32x4;
__device__ __forceinline__ i32x4 mk(const void* p) {
  unsigned long long a = (unsigned long long)p;
  i32x4 r;
  r[0] = __builtin_amdgcn_readfirstlane((int)(a & 0xffffffffu));
  r[1] = __builtin_amdgcn_readfirstlane((int)((a >> 32) & 0xffffu));
  r[2] = 0x7ffffff0; r[3] = 0x00020000;
  return r;
}
// every WG streams `iters` chunks of CHUNK bytes starting at its own offset (wrapping inside `bytes`)
template <int DEPTH>   // DEPTH 1 KiB-per-wave DMA instructions in flight per wave
__global__ __launch_bounds__(256) void dma_kernel(const char* base, size_t bytes, int iters, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // DEPTH * 4 KiB
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t wg_span = (size_t)DEPTH * 4096;
  size_t off = ((size_t)blockIdx.x * 1315423911ull) % (bytes / wg_span) * wg_span;
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0" : "=s"(keep));
  for (int it = 0; it < iters; ++it) {
    const char* p = base + off;
    i32x4 rv = mk(p);
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const unsigned ldsaddr = (unsigned)(size_t)lds + (d * 4 + wave) * 1024;
      const int voff = (d * 4 + wave) * 1024 + lane * 16;
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(__builtin_amdgcn_readfirstlane(ldsaddr)), "v"(voff), "s"(rv) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    off += wg_span * 977;
    if (off + wg_span > bytes) off %= (bytes - wg_span), off &= ~(size_t)4095;
  }
  asm volatile("s_mov_b32 m0, %0" ::"s"(keep));
  __syncthreads();
  if (threadIdx.x == 0) sink[blockIdx.x] = ((unsigned*)lds)[lane];
}
template <int DEPTH>   // DEPTH 16-byte loads in flight per thread
__global__ __launch_bounds__(256) void reg_kernel(const char* base, size_t bytes, int iters, unsigned* sink) {
  extern __shared__ unsigned char lds[];   // only to pin the occupancy
  const size_t wg_span = (size_t)DEPTH * 4096;
  size_t off = ((size_t)blockIdx.x * 1315423911ull) % (bytes / wg_span) * wg_span;
  int4 acc = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    const int4* p = reinterpret_cast<const int4*>(base + off) + threadIdx.x;
    int4 v[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) v[d] = p[d * 256];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) { acc.x ^= v[d].x; acc.y ^= v[d].y; acc.z ^= v[d].z; acc.w ^= v[d].w; }
    off += wg_span * 977;
    if (off + wg_span > bytes) off %= (bytes - wg_span), off &= ~(size_t)4095;
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345) sink[blockIdx.x] = acc.x;
  if (threadIdx.x == 0) lds[0] = 1;
}
template <typename F> static double run(F launch) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  launch(); hipDeviceSynchronize();
  hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms;
}
int main() {
  const size_t sizes[3] = {8ull << 20, 96ull << 20, 2048ull << 20};
  const char* names[3] = {"8 MiB (L2)", "96 MiB (MALL)", "2 GiB (HBM)"};
  char* buf; unsigned* sink;
  hipMalloc(&buf, sizes[2]); hipMemset(buf, 1, sizes[2]); hipMalloc(&sink, 1 << 20);
  const int cus = 256; const double clk = 2.4e9;
  for (int s = 0; s < 3; ++s) {
    for (int wgs_per_cu : {2, 5, 8}) {
      const int grid = cus * wgs_per_cu, iters = 400;
      const size_t lds_pin = (size_t)(160 * 1024 / wgs_per_cu) & ~(size_t)1023;   // exactly wgs_per_cu WGs fit
#define DMA(D) { if ((size_t)D * 4096 <= lds_pin) { hipFuncSetAttribute((const void*)dma_kernel<D>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_pin); \
      double ms = run([&] { hipLaunchKernelGGL(dma_kernel<D>, dim3(grid), dim3(256), lds_pin, 0, buf, sizes[s], iters, sink); }); \
      double bytes = (double)grid * iters * D * 4096; \
      printf("%-14s %d WG/CU  DMA in-flight %3d KiB/WG: %7.0f GB/s  %5.1f B/clk/CU\n", names[s], wgs_per_cu, D * 4, bytes / ms / 1e6, bytes / (ms * 1e-3) / clk / cus); } }
#define REG(D) { hipFuncSetAttribute((const void*)reg_kernel<D>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_pin); \
      double ms = run([&] { hipLaunchKernelGGL(reg_kernel<D>, dim3(grid), dim3(256), lds_pin, 0, buf, sizes[s], iters, sink); }); \
      double bytes = (double)grid * iters * D * 4096; \
      printf("%-14s %d WG/CU  REG in-flight %3d KiB/WG: %7.0f GB/s  %5.1f B/clk/CU\n", names[s], wgs_per_cu, D * 4, bytes / ms / 1e6, bytes / (ms * 1e-3) / clk / cus); }
      DMA(4) DMA(8) DMA(16)
      REG(2) REG(4) REG(8)
    }
  }
  return 0;
}
