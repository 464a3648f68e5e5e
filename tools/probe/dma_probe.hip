// Probe: semantics of `buffer_load_dwordx4 ... lds` (LDS-DMA) on gfx950 -- destination layout and
// what out-of-range lanes write.  Build+run on the GPU box: hipcc --offload-arch=gfx950 -O3 dma_probe.hip -o /tmp/dma_probe && /tmp/dma_probe
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
__global__ void k(const unsigned* p, unsigned* out) {
  __shared__ __attribute__((aligned(16))) unsigned lds[2048];
  for (int i = threadIdx.x; i < 2048; i += 256) lds[i] = 0xdeadbeefu;
  __syncthreads();
  i32x4 rv = mk(p);
  int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // lane l fetches global chunk (63 - l) of its wave's 1 KiB region; every 4th lane is out of range
  int voff = (lane % 4 == 3) ? (int)0x80000000u : (wave * 1024 + (63 - lane) * 16);
  unsigned ldsaddr = (unsigned)(size_t)lds + wave * 1024;
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "s"(__builtin_amdgcn_readfirstlane(ldsaddr)), "v"(voff), "s"(rv) : "memory");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  for (int i = threadIdx.x; i < 1024; i += 256) out[i] = lds[i];
}
int main() {
  std::vector<unsigned> h(1024);
  for (int i = 0; i < 1024; ++i) h[i] = 0x1000 + i;  // dword index
  unsigned *d, *o;
  hipMalloc(&d, 4096); hipMalloc(&o, 4096);
  hipMemcpy(d, h.data(), 4096, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, d, o);
  std::vector<unsigned> r(1024);
  hipMemcpy(r.data(), o, 4096, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int w = 0; w < 4; ++w) for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) {
    unsigned got = r[w * 256 + l * 4 + j];
    unsigned want = (l % 4 == 3) ? 0u : (0x1000 + w * 256 + (63 - l) * 4 + j);
    if (got != want) { if (bad < 8) printf("wave %d lane %d dw %d: got %08x want %08x\n", w, l, j, got, want); ++bad; }
  }
  printf("LDS-DMA probe: %s (%d mismatches); lane0: %08x %08x, lane3(OOB): %08x\n", bad ? "UNEXPECTED" : "as modelled (dst = M0 + lane*16, OOB -> 0)", bad, r[0], r[1], r[12]);
  return 0;
}
