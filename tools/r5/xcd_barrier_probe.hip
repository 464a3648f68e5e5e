// Probe: what does a barrier among the 32 workgroups of ONE XCD cost, and does a hand-off through that XCD's L2
// (plain stores -> s_waitcnt vmcnt(0) -> L2-scope atomic arrive -> L1-bypassing reads) deliver fresh data?
//
// Every workgroup reads its XCC id from the hardware register (a runtime FACT, not an assumption about the dispatcher), takes a ticket on
// that XCD's arrival counter and from then on talks only to workgroups of the same XCD.  All of them share one physical L2, so no L2
// write-back (`buffer_wbl2`) and no cross-XCD release is needed: a store is visible to every CU of the XCD once the L2 has acknowledged
// it (vmcnt), provided the reader does not hit a stale line of its own L1 (sc1 loads bypass it; `buffer_inv sc1` drops it).
//
// Variants timed (per phase, N phases inside one launch, device 100 MHz clock + host events):
//   0  barrier only                         arrive (atomic add, no sc1) + poll (sc1 load) on the XCD's counter
//   1  4 KB payload per WG + barrier + every WG reads the XCD's 128 KB back with sc1 loads            (checks every word)
//   2  the same with ONE `buffer_inv sc1` after the barrier and plain loads                          (checks every word)
//   3  barrier only, agent-scope atomics (sc1) for comparison
// Also prints the ticket census (workgroups per XCD) idle and beside a streaming kernel on a second stream.
// Build: hipcc --offload-arch=gfx950 -O3 tools/r5/xcd_barrier_probe.hip -o build/xcd_barrier_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef __attribute__((address_space(1))) unsigned gu32;

__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 15u;
}
__device__ __forceinline__ void l2_atomic_inc(unsigned* p) {          // no sc1: performed in this XCD's L2
    asm volatile("global_atomic_add %0, %1, off" ::"v"(p), "v"(1u) : "memory");
}
__device__ __forceinline__ unsigned l2_atomic_inc_ret(unsigned* p) {
    unsigned r;
    asm volatile("global_atomic_add %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(p), "v"(1u) : "memory");
    return r;
}
__device__ __forceinline__ void agent_atomic_inc(unsigned* p) {
    asm volatile("global_atomic_add %0, %1, off sc1" ::"v"(p), "v"(1u) : "memory");
}
__device__ __forceinline__ unsigned load_sc1(const unsigned* p) {     // bypasses the CU's L1, served by the L2
    unsigned r;
    asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(p) : "memory");
    return r;
}
__device__ __forceinline__ uint4 load4_sc1(const uint4* p) {
    uint4 r;
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(r) : "v"(p) : "memory");
    return r;
}

struct Ctl {
    unsigned arrive[8][32];     // one counter per XCD, 128 bytes apart
    unsigned ticket[8][32];
    unsigned timeout;
    unsigned census[8];
};

// XCD-local barrier: `target` arrivals expected on arrive[x]
template <bool AGENT>
__device__ __forceinline__ bool xcd_barrier(Ctl* c, unsigned x, unsigned target, bool& dead) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's stores are in the L2
    __syncthreads();
    if (threadIdx.x == 0 && !dead) {
        if (AGENT) agent_atomic_inc(&c->arrive[x][0]); else l2_atomic_inc(&c->arrive[x][0]);
        unsigned spins = 0;
        while (load_sc1(&c->arrive[x][0]) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > 400000u) { atomicAdd(&c->timeout, 1u); dead = true; break; }
        }
    }
    __syncthreads();
    return !dead;
}

template <int VARIANT>
__global__ __launch_bounds__(256) void probe_kernel(Ctl* c, unsigned* payload, int phases, unsigned long long* wg_clk, unsigned* errs, unsigned* where) {
    extern __shared__ unsigned char lds_pin[];            // 96 KB: one workgroup per CU
    __shared__ unsigned s_x, s_slot;
    __shared__ bool s_dead;
    if (threadIdx.x == 0) {
        const unsigned x = xcc_id();
        s_x = x;
        s_slot = l2_atomic_inc_ret(&c->ticket[x][0]);
        s_dead = false;
        where[blockIdx.x] = x * 1000u + s_slot;
    }
    __syncthreads();
    const unsigned x = s_x, slot = s_slot;
    bool dead = false;
    unsigned* const mine = payload + (size_t)x * (32 * 1024) + slot * 1024;      // 4 KB per workgroup, 128 KB per XCD
    const unsigned* const grp = payload + (size_t)x * (32 * 1024);
    unsigned bad = 0;
    // first barrier: everybody of this XCD has arrived (census) -- not timed
    unsigned epoch = 0;
    xcd_barrier<VARIANT == 3>(c, x, 32u * ++epoch, dead);
    const unsigned long long t0 = wall_clock64();
    for (int ph = 0; ph < phases; ++ph) {
        if (VARIANT == 1 || VARIANT == 2) {
            // payload: word i of my 4 KB = f(phase, slot, i); plain 16-byte stores
            const unsigned base = (unsigned)ph * 0x9E3779B1u + slot * 0x10001u;
            uint4 v = make_uint4(base + threadIdx.x * 4, base + threadIdx.x * 4 + 1, base + threadIdx.x * 4 + 2, base + threadIdx.x * 4 + 3);
            *reinterpret_cast<uint4*>(mine + threadIdx.x * 4) = v;
        }
        xcd_barrier<VARIANT == 3>(c, x, 32u * ++epoch, dead);
        if (VARIANT == 1 || VARIANT == 2) {
            if (VARIANT == 2) asm volatile("buffer_inv sc1" ::: "memory");
            // read all 32 slots of the XCD: 128 KB = 8192 uint4, 32 per thread, 8 in flight
            for (int it = 0; it < 32; it += 8) {
                uint4 r[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const uint4* p = reinterpret_cast<const uint4*>(grp) + (it + u) * 256 + threadIdx.x;
                    if (VARIANT == 1) r[u] = load4_sc1(p); else r[u] = *p;
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const unsigned s = (unsigned)(it + u);
                    const unsigned base = (unsigned)ph * 0x9E3779B1u + s * 0x10001u;
                    bad += (r[u].x != base + threadIdx.x * 4) + (r[u].y != base + threadIdx.x * 4 + 1) + (r[u].z != base + threadIdx.x * 4 + 2) +
                           (r[u].w != base + threadIdx.x * 4 + 3);
                }
            }
            // second barrier of the phase: nobody overwrites a slot that is still being read
            xcd_barrier<false>(c, x, 32u * ++epoch, dead);
        }
        if (dead) break;
    }
    const unsigned long long t1 = wall_clock64();
    if (threadIdx.x == 0) wg_clk[blockIdx.x] = t1 - t0;
    if (bad) atomicAdd(errs, bad);
    if (threadIdx.x == 0) lds_pin[0] = 1;
}

__global__ __launch_bounds__(256) void stream_kernel(const uint4* src, size_t n, unsigned* sink, int reps) {
    uint4 a = make_uint4(0, 0, 0, 0);
    for (int r = 0; r < reps; ++r)
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { const uint4 v = src[i]; a.x ^= v.x; a.y ^= v.y; a.z ^= v.z; a.w ^= v.w; }
    if ((a.x ^ a.y ^ a.z ^ a.w) == 0x1234567) sink[0] = a.x;
}

template <int VARIANT>
static void run(const char* name, Ctl* c, unsigned* payload, unsigned long long* clk, unsigned* errs, unsigned* where, int phases, bool loaded, const uint4* big, size_t nbig,
                unsigned* sink) {
    hipStream_t s1, s2;
    CK(hipStreamCreate(&s1));
    CK(hipStreamCreate(&s2));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipFuncSetAttribute((const void*)probe_kernel<VARIANT>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    float best = 1e9f;
    std::vector<unsigned long long> h(256);
    std::vector<unsigned> hw(256);
    unsigned herr = 0, hto = 0;
    Ctl hc;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipMemsetAsync(c, 0, sizeof(Ctl), s1));
        CK(hipMemsetAsync(errs, 0, 4, s1));
        CK(hipStreamSynchronize(s1));
        if (loaded) hipLaunchKernelGGL(stream_kernel, dim3(512), dim3(256), 0, s2, big, nbig, sink, 3);
        CK(hipEventRecord(e0, s1));
        hipLaunchKernelGGL((probe_kernel<VARIANT>), dim3(256), dim3(256), 96 * 1024, s1, c, payload, phases, clk, errs, where);
        CK(hipEventRecord(e1, s1));
        CK(hipStreamSynchronize(s1));
        CK(hipStreamSynchronize(s2));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms);
    }
    CK(hipMemcpy(h.data(), clk, 256 * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hw.data(), where, 256 * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(&herr, errs, 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(&hc, c, sizeof(Ctl), hipMemcpyDeviceToHost));
    hto = hc.timeout;
    unsigned long long mn = ~0ull, mx = 0;
    for (auto v : h) { mn = std::min(mn, v); mx = std::max(mx, v); }
    int census[8] = {0, 0, 0, 0, 0, 0, 0, 0}, rr = 0;
    for (int b = 0; b < 256; ++b) { census[(hw[b] / 1000) & 7]++; rr += ((hw[b] / 1000) == (unsigned)(b % 8)); }
    printf("%-46s %s  host %.2f us/phase | device clock min %.2f max %.2f us/phase | wrong words %u | timeouts %u | census %d %d %d %d %d %d %d %d | block b on XCD b%%8: %d/256\n",
           name, loaded ? "loaded" : "idle  ", best * 1000.f / phases, mn * 0.01 / phases, mx * 0.01 / phases, herr, hto, census[0], census[1], census[2], census[3], census[4],
           census[5], census[6], census[7], rr);
}

int main() {
    Ctl* c;
    unsigned *payload, *errs, *where, *sink;
    unsigned long long* clk;
    uint4* big;
    const size_t nbig = (size_t)1 << 26;     // 1 GiB
    CK(hipMalloc(&c, sizeof(Ctl)));
    CK(hipMalloc(&payload, 8 * 32 * 4096));
    CK(hipMalloc(&errs, 4));
    CK(hipMalloc(&where, 256 * 4));
    CK(hipMalloc(&sink, 4));
    CK(hipMalloc(&clk, 256 * 8));
    CK(hipMalloc(&big, nbig * 16));
    CK(hipMemset(big, 1, nbig * 16));
    const int phases = 200;
    for (int loaded = 0; loaded < 2; ++loaded) {
        run<0>("0 barrier only (L2-scope atomics)", c, payload, clk, errs, where, phases, loaded, big, nbig, sink);
        run<3>("3 barrier only (agent-scope atomics)", c, payload, clk, errs, where, phases, loaded, big, nbig, sink);
        run<1>("1 4 KB/WG + 2 barriers + 128 KB sc1 read-back", c, payload, clk, errs, where, phases, loaded, big, nbig, sink);
        run<2>("2 4 KB/WG + 2 barriers + buffer_inv + plain", c, payload, clk, errs, where, phases, loaded, big, nbig, sink);
    }
    return 0;
}
