// LDS atomic throughput probe (gfx950): cycles per wave-instruction for ds_add_f32 / ds_add_u32 / plain RMW, distinct addresses
// per lane vs the two wave halves sharing addresses.   hipcc --offload-arch=gfx950 -O3 -o lds_atomic lds_atomic.hip
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE, int SHARE>
__global__ __launch_bounds__(256) void k(float* out, long long* cyc, int iters) {
    __shared__ __attribute__((aligned(16))) float s[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) s[i] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int a0 = SHARE ? (lane & 31) : lane;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int idx = a0 + u * 64 + ((it & 3) << 10);
            if (MODE == 0) atomicAdd(&s[idx], 1.0f);
            else if (MODE == 1) atomicAdd(reinterpret_cast<unsigned*>(&s[idx]), 1u);
            else if (MODE == 2) s[idx] += 1.0f;
            else if (MODE == 3) atomicAdd(reinterpret_cast<int*>(&s[idx]), (int)lane);
            else if (MODE == 4) atomicAdd(reinterpret_cast<double*>(s) + idx, 1.0);
            else if (MODE == 5) atomicAdd(reinterpret_cast<unsigned long long*>(s) + idx, (unsigned long long)lane);
            else if (MODE == 6) out[idx] = atomicAdd(&s[idx], 1.0f);
            else if (MODE == 7) __hip_atomic_fetch_add(&s[idx], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    __syncthreads();
    long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    out[blockIdx.x * 256 + threadIdx.x] = s[threadIdx.x];
}
template <int MODE, int SHARE> void run(const char* name, float* out, long long* cyc, int wgs_per_cu) {
    const int iters = 256;
    hipLaunchKernelGGL((k<MODE, SHARE>), dim3(256 * wgs_per_cu), dim3(256), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); hipLaunchKernelGGL((k<MODE, SHARE>), dim3(256 * wgs_per_cu), dim3(256), 0, 0, out, cyc, iters); hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_cu = (double)iters * 16 * 4 * wgs_per_cu;
    printf("%-28s wgs/cu %d: %8.1f us  -> %6.1f ns per wave-instruction per CU\n", name, wgs_per_cu, ms * 1e3, ms * 1e6 / instr_per_cu);
}
int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 8 * 256 * 4); hipMalloc(&cyc, 256 * 8 * 8);
    for (int w = 1; w <= 2; ++w) {
        run<0, 0>("ds_add_f32 distinct", out, cyc, w);
        run<0, 1>("ds_add_f32 halves share", out, cyc, w);
        run<1, 0>("ds_add_u32 distinct", out, cyc, w);
        run<1, 1>("ds_add_u32 halves share", out, cyc, w);
        run<2, 0>("plain rmw distinct", out, cyc, w);
        run<3, 1>("ds_add_i32 share", out, cyc, w);
        run<4, 0>("ds_add_f64 distinct", out, cyc, w);
        run<5, 0>("ds_add_u64 distinct", out, cyc, w);
        run<5, 1>("ds_add_u64 halves share", out, cyc, w);
        run<6, 0>("ds_add_rtn_f32 distinct", out, cyc, w);
        run<7, 0>("f32 workgroup-scope relaxed", out, cyc, w);
    }
    return 0;
}
