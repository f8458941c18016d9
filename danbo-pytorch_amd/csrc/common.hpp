// Shared host/device helpers for the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/danbo_hip.h"
#include "sample_math.hpp"

#define DANBO_CHECK_ARG(cond) do { if (!(cond)) return DANBO_EINVAL; } while (0)
#define DANBO_LAUNCH_RET() do { hipError_t e_ = hipGetLastError(); return e_ == hipSuccess ? 0 : (int)e_; } while (0)

namespace danbo {

constexpr int WAVE = 64;
constexpr int NUM_CU = 256;  // MI355X; grids of persistent kernels are sized from this

static inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

// grid for a memory-bound grid-stride kernel: enough workgroups to fill 256 CUs x 8
static inline int stream_grid(long items, int block) {
    long g = (items + block - 1) / block;
    const long cap = (long)NUM_CU * 8;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// number of rows a kernel has to process: device-side count (clamped to capacity) or host n
__device__ __forceinline__ int resolve_count(const int32_t* count, int n_cap) {
    if (count == nullptr) return n_cap;
    int c = *count;
    return c < n_cap ? c : n_cap;
}

}  // namespace danbo
