// Calibration probe (dev tool): what does the MLP16 instruction mix cost on this GPU?
//   hipcc --offload-arch=gfx950 -O3 -o mfma_probe mfma_probe.hip && ./mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int CHUNK = 32768, SLOTS = 4, NCH = 74;

// FLAGS: 1 = ds_read A fragments, 2 = barrier per chunk, 4 = global_load_lds stream, 8 = VALU filler, 16 = 32x32x16 shape
template <int FLAGS>
__global__ __launch_bounds__((FLAGS & 32) ? 256 : 512, (FLAGS & 32) ? 1 : 2) void k_probe(const char* packed, float* out, int iters, long long* clk) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < SLOTS * CHUNK / 4; i += blockDim.x) reinterpret_cast<float*>(smem)[i] = 0.001f * (i & 255);
    __syncthreads();
    const long long c0 = clock64(), w0 = wall_clock64();
    half8 bh, bl;
    for (int e = 0; e < 8; ++e) { bh[e] = (_Float16)(0.01f * (lane + e)); bl[e] = (_Float16)(0.0001f * (lane - e)); }
    f32x4 acc[16];
    for (int T = 0; T < 16; ++T) acc[T] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x16 big[4];
    f32x16 wide[8];
    for (int T = 0; T < 8; ++T) for (int e = 0; e < 16; ++e) wide[T][e] = 0.f;
    for (int T = 0; T < 4; ++T) for (int e = 0; e < 16; ++e) big[T][e] = 0.f;
    float filler[8];
    for (int e = 0; e < 8; ++e) filler[e] = lane * 0.5f + e;
    int slot = 0, chunk = 0, islot = 3 % SLOTS, ichunk = 3;
    for (int it = 0; it < iters; ++it) {
        if (FLAGS & 2) {
            if (FLAGS & 4) { if (FLAGS & 32) asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
            __builtin_amdgcn_s_barrier();
        }
        if (FLAGS & 4) {
            constexpr int PER = (FLAGS & 32) ? 8 : 4;
            const char* src = packed + (size_t)ichunk * CHUNK + wave * (PER * 1024) + lane * 16;
            char* dst = smem + islot * CHUNK + wave * (PER * 1024);
#pragma unroll
            for (int q = 0; q < PER; ++q)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + q * 1024),
                                                 (__attribute__((address_space(3))) void*)(dst + q * 1024), 16, 0, 0);
            ichunk = ichunk + 1 == NCH ? 0 : ichunk + 1;
            islot = islot + 1 == SLOTS ? 0 : islot + 1;
        }
        const char* base = smem + slot * CHUNK + lane * 16;
        slot = slot + 1 == SLOTS ? 0 : slot + 1;
        if (FLAGS & 8) {
#pragma unroll
            for (int r = 0; r < ((FLAGS & 32) ? 18 : 9); ++r)
#pragma unroll
                for (int e = 0; e < 8; ++e) filler[e] = fmaxf(filler[e] * 1.0001f + 0.5f, 0.25f);
        }
        if (FLAGS & 32) {
            f32x16 (&a8)[8] = *reinterpret_cast<f32x16 (*)[8]>(&wide[0]);
#pragma unroll
            for (int b = 0; b < 16; ++b) {   // 2 k-steps x 8 tiles of 32 features x 32 samples
                half8 ah, al;
                if (FLAGS & 1) {
                    ah = *reinterpret_cast<const half8*>(base + (2 * b) * 1024);
                    al = *reinterpret_cast<const half8*>(base + (2 * b + 1) * 1024);
                } else { ah = bh; al = bl; }
                a8[b & 7] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, a8[b & 7], 0, 0, 0);
                a8[b & 7] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, a8[b & 7], 0, 0, 0);
                a8[b & 7] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, a8[b & 7], 0, 0, 0);
            }
        } else if (FLAGS & 16) {
#pragma unroll
            for (int b = 0; b < 4; ++b) {   // 4 x (32 features x 32 samples): 4 tiles x 3 x 2 k-halves = 24 MFMAs of 32 cycles
                half8 ah[2], al[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    if (FLAGS & 1) {
                        ah[t] = *reinterpret_cast<const half8*>(base + (8 * b + 4 * t) * 1024);
                        al[t] = *reinterpret_cast<const half8*>(base + (8 * b + 4 * t + 1) * 1024);
                    } else { ah[t] = bh; al[t] = bl; }
                }
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    big[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[t], bh, big[b], 0, 0, 0);
                    big[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[t], bl, big[b], 0, 0, 0);
                    big[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[t], bh, big[b], 0, 0, 0);
                }
            }
        } else {
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                half8 ah[2], al[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    if (FLAGS & 1) {
                        ah[t] = *reinterpret_cast<const half8*>(base + (4 * b + 2 * t) * 1024);
                        al[t] = *reinterpret_cast<const half8*>(base + (4 * b + 2 * t + 1) * 1024);
                    } else { ah[t] = bh; al[t] = bl; }
                }
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    acc[2 * b + t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[t], bh, acc[2 * b + t], 0, 0, 0);
                    acc[2 * b + t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[t], bl, acc[2 * b + t], 0, 0, 0);
                    acc[2 * b + t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[t], bh, acc[2 * b + t], 0, 0, 0);
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    float s = 0.f;
    for (int T = 0; T < 16; ++T) s += acc[T][0] + acc[T][1] + acc[T][2] + acc[T][3];
    for (int T = 0; T < 4; ++T) for (int e = 0; e < 16; ++e) s += big[T][e];
    for (int e = 0; e < 8; ++e) s += filler[e];
    for (int T = 0; T < 8; ++T) for (int e = 0; e < 16; ++e) s += wide[T][e];
    out[blockIdx.x * blockDim.x + tid] = s;
    if (tid == 0 && blockIdx.x == 0) { clk[0] = clock64() - c0; clk[1] = wall_clock64() - w0; }
}

template <int FLAGS>
void run(const char* name, const char* packed, float* out, int iters) {
    static long long* clk = nullptr; if (!clk) hipHostMalloc(&clk, 16);
    const int lds = SLOTS * CHUNK;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_probe<FLAGS>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_probe<FLAGS>, dim3(256), dim3((FLAGS & 32) ? 256 : 512), lds, 0, packed, out, iters, clk);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k_probe<FLAGS>, dim3(256), dim3((FLAGS & 32) ? 256 : 512), lds, 0, packed, out, iters, clk);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
    // flops per iteration per wave: 48 MFMA x 16384 (16x16x32) or 24 x 32768 (32x32x16)
    const double flop = (double)iters * 256 * 8 * 48 * 16384.0;   // the 4-wave mode does 4 x 48 x 32768: the same
    printf("%-44s %8.3f ms  %7.1f TFLOP/s (f16 MFMA)  -> MLP16-equivalent %6.1f of 833\n", name, ms, flop / ms / 1e9, flop / ms / 1e9 / 3);
    printf("      s_memtime ticks %lld, 100MHz ticks %lld -> %.0f MHz if s_memtime is the shader clock\n", clk[0], clk[1], (double)clk[0] / clk[1] * 100.0);
}

int main() {
    char* packed; float* out;
    hipMalloc(&packed, (size_t)NCH * CHUNK);
    hipMemset(packed, 0, (size_t)NCH * CHUNK);
    hipMalloc(&out, 256 * 512 * 4);
    const int iters = 74 * 20;   // 20 row tiles per workgroup
    run<0>("mfma 16x16x32 only", packed, out, iters);
    run<16>("mfma 32x32x16 only", packed, out, iters);
    run<1>("16x16x32 + ds_read A", packed, out, iters);
    run<1 | 16>("32x32x16 + ds_read A", packed, out, iters);
    run<1 | 2>("16x16x32 + ds_read + barrier", packed, out, iters);
    run<1 | 2 | 4>("16x16x32 + ds_read + barrier + stream", packed, out, iters);
    run<1 | 2 | 4 | 8>("16x16x32 + ds_read + barrier + stream + valu", packed, out, iters);
    run<1 | 8>("16x16x32 + ds_read + valu", packed, out, iters);
    run<8>("16x16x32 + valu", packed, out, iters);
    run<1 | 2 | 4 | 8 | 16>("32x32x16 + ds_read + barrier + stream + valu", packed, out, iters);
    run<32 | 16>("1 wave/SIMD 32x32x16 x 32 samples: mfma only", packed, out, iters);
    run<32 | 16 | 1>("   + ds_read", packed, out, iters);
    run<32 | 16 | 1 | 2 | 4>("   + ds_read + barrier + stream", packed, out, iters);
    run<32 | 16 | 1 | 2 | 4 | 8>("   + ds_read + barrier + stream + valu", packed, out, iters);
    return 0;
}
