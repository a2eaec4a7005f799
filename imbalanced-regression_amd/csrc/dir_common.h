// Shared helpers for the libdir_hip.so translation units (gfx950 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include "dir_hip.h"

#define DIR_WAVE 64
#define DIR_TPB 256                     // 4 waves: one per SIMD of a CU

#define DIR_RETURN_IF(cond, code) do { if (cond) return (code); } while (0)
#define DIR_LAUNCH_CHECK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return (int)e_; } while (0)

// One-time per-DEVICE setup of a launch site (hipFuncSetAttribute for kernels with > 64 KB of dynamic LDS): a process may drive
// several GPUs (the attribute is per device), so the "done" flag is a bit per device ordinal, not a process-wide bool. The setup runs
// BEFORE the bit is published (release), so a second thread that sees the bit (acquire) also sees a configured function; two threads
// racing through the first use both make the (idempotent) calls.
static inline uint64_t dir_device_setup_pending(std::atomic<uint64_t>& done) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    const uint64_t bit = 1ull << (dev & 63);
    return (done.load(std::memory_order_acquire) & bit) ? 0 : bit;
}
#define DIR_ONCE_PER_DEVICE(...) do { static std::atomic<uint64_t> done_{0}; const uint64_t bit_ = dir_device_setup_pending(done_); \
                                      if (bit_) { __VA_ARGS__; done_.fetch_or(bit_, std::memory_order_release); } } while (0)

static inline hipStream_t dir_s(dir_stream_t s) { return reinterpret_cast<hipStream_t>(s); }
static inline bool dir_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static inline size_t dir_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
static inline int dir_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// The reference's per-label branch (imdb-wiki-dir/fds.py:91-99 / :120-137) as a per-row rule.
// lo = (float)bucket_start, hi = (float)(bucket_num - 1). Returns the table row or -1.
__device__ __forceinline__ int dir_bin_of(float l, float lo, float hi, bool has_lo, bool has_hi) {
    if (l != l) return -1;                       // NaN: matches no row group
    if (l > hi) return has_hi ? (int)(hi - lo) : -1;
    if (l < lo) return has_lo ? 0 : -1;
    return (int)(l - lo);                        // float32 subtract, truncate (fds.py:104)
}

__device__ __forceinline__ uint32_t dir_label_flag_bits(float l, float lo, float hi) {
    uint32_t f = 0;
    if (l != l) return DIR_FLAG_NAN;
    if (l == lo) f |= DIR_FLAG_HAS_LO;
    if (l == hi) f |= DIR_FLAG_HAS_HI;
    if (l >= lo && l <= hi && l != truncf(l)) f |= DIR_FLAG_NONINTEGER;
    return f;
}

// 64-lane butterfly reductions (wavefront shuffles; no LDS).
__device__ __forceinline__ double dir_wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, DIR_WAVE);
    return v;
}
__device__ __forceinline__ uint32_t dir_wave_or(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v |= (uint32_t)__shfl_xor((int)v, o, DIR_WAVE);
    return v;
}
