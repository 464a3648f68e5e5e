// Shared device/host helpers for the TOIST hot-path kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <cstdlib>

#include "../../include/toist_hip.h"

typedef unsigned short bf16_t;  // raw bfloat16 bits
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;

namespace toist {

// ---- error plumbing (host) -------------------------------------------------
void set_last_error(const char* fmt, ...);
int check_launch(const char* what);  // hipGetLastError -> TOIST_EHIP / TOIST_OK

#define TOIST_REQUIRE(cond, ...)                  \
    do {                                          \
        if (!(cond)) {                            \
            ::toist::set_last_error(__VA_ARGS__); \
            return TOIST_EINVAL;                  \
        }                                         \
    } while (0)

// ---- tuning switches --------------------------------------------------------------------
// The dispatcher's thresholds were found with environment switches (tools/dbg/ab_*.sh).  A product build compiles them to their
// defaults; `make KNOBS=1` (-DTOIST_TUNING_KNOBS) reads the environment once per process for A/B runs.
#ifdef TOIST_TUNING_KNOBS
inline long long tuning_knob(const char* name, long long dflt) {
    const char* e = std::getenv(name);
    return e ? std::atoll(e) : dflt;
}
#else
inline long long tuning_knob(const char*, long long dflt) { return dflt; }
#endif

// ---- per-device one-time kernel attributes (> 64 KiB of dynamic LDS) ----------------------
// hipFuncSetAttribute is per device; a process that drives several devices must set it on each.  Lock-free and idempotent: two
// threads racing on the first launch both set the same value.  `done` = one bit per device ordinal.
template <typename F>
inline bool lds_attr_once_flag(std::atomic<unsigned long long>& done, F&& set) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return true;
    if (!set()) return false;
    done.fetch_or(bit, std::memory_order_release);
    return true;
}
template <typename F>
inline bool lds_attr_once(int slot, F&& set) {
    static std::atomic<unsigned long long> done[8];
    return lds_attr_once_flag(done[slot & 7], static_cast<F&&>(set));
}

// ---- bf16 <-> f32 ------------------------------------------------------------
__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((unsigned)h) << 16); }

// float -> bfloat16, round-to-nearest-even (torch's cast): gfx950 converts in hardware, two values per
// v_cvt_pk_bf16_f32 -- the hand-rolled rounding was 5-6 VALU operations per value in every epilogue
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_native_t;

__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }

__device__ __forceinline__ unsigned pack2bf(float lo, float hi) {
    const bf16x2_native_t v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(unsigned, v);
}

// ---- wave (64-lane) reductions -------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// counter-based RNG for dropout: one 32-bit hash per (seed, element index).
// Forward and backward regenerate the same keep-mask from (seed, index).
__device__ __forceinline__ unsigned hash_u32(unsigned long long seed, unsigned long long idx) {
    // two-round 32-bit mixer (multiply / xor-shift, full avalanche) over the element index, keyed by the 64-bit seed.  The previous
    // splitmix64 finaliser cost ~25 VALU operations per element in 64-bit multiplies -- visible in every kernel that drops out.
    unsigned x = (unsigned)idx ^ (unsigned)seed;
    // the seed enters twice (xor before round 1, a multiplied copy before round 2): masks of different seeds are not index-shifted copies
    const unsigned key = ((unsigned)(seed >> 32) ^ ((unsigned)seed * 0x9E3779B9u)) + (unsigned)(idx >> 32) * 0x85EBCA6Bu;
    x ^= x >> 16; x *= 0x7feb352du;
    x ^= key;
    x ^= x >> 15; x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}
// keep with probability 1-p ; thresh = (unsigned)(p * 2^32)
__device__ __forceinline__ bool dropout_keep(unsigned long long seed, unsigned long long idx, unsigned thresh) {
    return hash_u32(seed, idx) >= thresh;
}

}  // namespace toist
