// Scheduling probe for k_pe_mlp16 (dev tool): the kernel's per-chunk instruction mix (32 ds_read_b128 of A
// fragments, 48 fp16 MFMAs, a 40-op bias/ReLU/split epilogue, ring hand-over) under different MFMA orders
// and scheduler fences.   hipcc --offload-arch=gfx950 -O3 -o sched_probe sched_probe.hip && ./sched_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int CHUNK = 32768, SLOTS = 4, NCH = 74;

__device__ __forceinline__ void split8(const float* v, half8& hi, half8& lo) {
#pragma unroll
    for (int e = 0; e < 8; ++e) { const _Float16 h = (_Float16)v[e]; hi[e] = h; lo[e] = (_Float16)(v[e] - (float)h); }
}
#define MF(acc, a, b) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0)

// ORDER: 0 = three dependent MFMAs per tile back to back; 1 = two tiles interleaved; 2 = four tiles: hh x4, hl x4, lh x4
// FENCE: 0 = sched_barrier around every 2-tile batch (the kernel today); 1 = none; 2 = sched_group_barrier MFMA/VALU/DS mix
template <int ORDER, int FENCE, bool EPI>
__global__ __launch_bounds__(512, 2) void k_probe(const char* packed, const float* bias, float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < SLOTS * CHUNK / 4; i += 512) reinterpret_cast<float*>(smem)[i] = 1e-3f * (i & 255);
    float* s_bias = reinterpret_cast<float*>(smem + SLOTS * CHUNK);
    if (tid < 256) s_bias[tid] = bias[tid];
    __syncthreads();
    f32x4 acc[16], prev[16];
    for (int T = 0; T < 16; ++T) { acc[T] = f32x4{0, 0, 0, 0}; prev[T] = f32x4{0.1f * lane, 0.2f, -0.3f, 0.4f * T}; }
    int slot = 0, islot = 3, ichunk = 3;
    const bool early = wave < 4;
    for (int it0 = 0; it0 < iters; it0 += 8) {
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        // epilogue piece: B fragment of this k-step from the previous layer's accumulators
        half8 bh, bl;
        {
            float v[8];
            const float4 b0 = *reinterpret_cast<const float4*>(s_bias + 32 * s + 4 * (lane >> 4));
            const float4 b1 = *reinterpret_cast<const float4*>(s_bias + 32 * s + 16 + 4 * (lane >> 4));
            if (EPI) {
                v[0] = fmaxf(prev[2 * s][0] + b0.x, 0.f); v[1] = fmaxf(prev[2 * s][1] + b0.y, 0.f);
                v[2] = fmaxf(prev[2 * s][2] + b0.z, 0.f); v[3] = fmaxf(prev[2 * s][3] + b0.w, 0.f);
                v[4] = fmaxf(prev[2 * s + 1][0] + b1.x, 0.f); v[5] = fmaxf(prev[2 * s + 1][1] + b1.y, 0.f);
                v[6] = fmaxf(prev[2 * s + 1][2] + b1.z, 0.f); v[7] = fmaxf(prev[2 * s + 1][3] + b1.w, 0.f);
            } else {
                for (int e = 0; e < 8; ++e) v[e] = b0.x + e;
            }
            split8(v, bh, bl);
        }
        auto handover = [&]() {
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const char* src = packed + (size_t)ichunk * CHUNK + wave * 4096 + lane * 16;
            char* dst = smem + islot * CHUNK + wave * 4096;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + q * 1024),
                                                 (__attribute__((address_space(3))) void*)(dst + q * 1024), 16, 0, 0);
            ichunk = ichunk + 1 == NCH ? 0 : ichunk + 1;
            islot = islot + 1 == SLOTS ? 0 : islot + 1;
        };
        if (!early) handover();
        const char* base = smem + slot * CHUNK + lane * 16;
        slot = slot + 1 == SLOTS ? 0 : slot + 1;
        auto frag = [&](int piece) { return *reinterpret_cast<const half8*>(base + piece * 1024); };
        if (ORDER == 4 || ORDER == 5) {
#define MFI(acc, a, b) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
            half8 ah[2][2], al[2][2];
#pragma unroll
            for (int t = 0; t < 2; ++t) { ah[0][t] = frag(2 * t); al[0][t] = frag(2 * t + 1); }
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                if (b + 1 < 8) {
#pragma unroll
                    for (int t = 0; t < 2; ++t) { ah[(b + 1) & 1][t] = frag(4 * (b + 1) + 2 * t); al[(b + 1) & 1][t] = frag(4 * (b + 1) + 2 * t + 1); }
                }
                if (b == 4 && early) handover();
                f32x4& a0 = acc[2 * b];
                f32x4& a1 = acc[2 * b + 1];
                asm volatile("s_nop 1");
                if (ORDER == 4) {
                    MFI(a0, ah[b & 1][0], bh); MFI(a0, ah[b & 1][0], bl); MFI(a0, al[b & 1][0], bh);
                    MFI(a1, ah[b & 1][1], bh); MFI(a1, ah[b & 1][1], bl); MFI(a1, al[b & 1][1], bh);
                } else {
                    MFI(a0, ah[b & 1][0], bh); MFI(a1, ah[b & 1][1], bh);
                    MFI(a0, ah[b & 1][0], bl); MFI(a1, ah[b & 1][1], bl);
                    MFI(a0, al[b & 1][0], bh); MFI(a1, al[b & 1][1], bh);
                }
            }
            if (s == 7) asm volatile("s_nop 7\n s_nop 7");
        } else if (ORDER == 3) {
            half8 ah[2][2], al[2][2];
            const unsigned laddr = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)base;
            auto rd = [&](half8& dst, int piece) {
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(laddr + (piece >= 32 ? 0 : 0)), "n"(0) : "memory");
            };
            (void)rd;
#define RD(dst, piece) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(laddr), "i"((piece) * 1024) : "memory")
            RD(ah[0][0], 0); RD(al[0][0], 1); RD(ah[0][1], 2); RD(al[0][1], 3);
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                if (b + 1 < 8) {
                    RD(ah[(b + 1) & 1][0], 4 * (b + 1)); RD(al[(b + 1) & 1][0], 4 * (b + 1) + 1);
                    RD(ah[(b + 1) & 1][1], 4 * (b + 1) + 2); RD(al[(b + 1) & 1][1], 4 * (b + 1) + 3);
                }
                if (b == 4 && early) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); handover(); }
                if (b + 1 < 8 && !(b == 4 && early))
                    asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(ah[b & 1][0]), "+v"(al[b & 1][0]), "+v"(ah[b & 1][1]), "+v"(al[b & 1][1]));
                else
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ah[b & 1][0]), "+v"(al[b & 1][0]), "+v"(ah[b & 1][1]), "+v"(al[b & 1][1]));
                f32x4& a0 = acc[2 * b];
                f32x4& a1 = acc[2 * b + 1];
                MF(a0, ah[b & 1][0], bh); MF(a1, ah[b & 1][1], bh);
                MF(a0, ah[b & 1][0], bl); MF(a1, ah[b & 1][1], bl);
                MF(a0, al[b & 1][0], bh); MF(a1, al[b & 1][1], bh);
                if (FENCE == 0) __builtin_amdgcn_sched_barrier(0);
            }
        } else if (ORDER == 2) {
            half8 ah[2][4], al[2][4];
#pragma unroll
            for (int t = 0; t < 4; ++t) { ah[0][t] = frag(2 * t); al[0][t] = frag(2 * t + 1); }
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                if (b + 1 < 4) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) { ah[(b + 1) & 1][t] = frag(8 * (b + 1) + 2 * t); al[(b + 1) & 1][t] = frag(8 * (b + 1) + 2 * t + 1); }
                }
                if (b == 2 && early) handover();
                if (FENCE == 0) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < 4; ++t) MF(acc[4 * b + t], ah[b & 1][t], bh);
#pragma unroll
                for (int t = 0; t < 4; ++t) MF(acc[4 * b + t], ah[b & 1][t], bl);
#pragma unroll
                for (int t = 0; t < 4; ++t) MF(acc[4 * b + t], al[b & 1][t], bh);
                if (FENCE == 0) __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            half8 ah[2][2], al[2][2];
#pragma unroll
            for (int t = 0; t < 2; ++t) { ah[0][t] = frag(2 * t); al[0][t] = frag(2 * t + 1); }
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                if (b + 1 < 8) {
#pragma unroll
                    for (int t = 0; t < 2; ++t) { ah[(b + 1) & 1][t] = frag(4 * (b + 1) + 2 * t); al[(b + 1) & 1][t] = frag(4 * (b + 1) + 2 * t + 1); }
                }
                if (b == 4 && early) handover();
                if (FENCE == 0) __builtin_amdgcn_sched_barrier(0);
                f32x4& a0 = acc[2 * b];
                f32x4& a1 = acc[2 * b + 1];
                if (ORDER == 0) {
                    MF(a0, ah[b & 1][0], bh); MF(a0, ah[b & 1][0], bl); MF(a0, al[b & 1][0], bh);
                    MF(a1, ah[b & 1][1], bh); MF(a1, ah[b & 1][1], bl); MF(a1, al[b & 1][1], bh);
                } else {
                    MF(a0, ah[b & 1][0], bh); MF(a1, ah[b & 1][1], bh);
                    MF(a0, ah[b & 1][0], bl); MF(a1, ah[b & 1][1], bl);
                    MF(a0, al[b & 1][0], bh); MF(a1, al[b & 1][1], bh);
                }
                if (FENCE == 0) __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (FENCE == 2) {
            // one region per chunk: after every MFMA allow one VALU, every third MFMA one DS read
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            }
        }
    }
#pragma unroll
        for (int T = 0; T < 16; ++T) { prev[T] = acc[T]; acc[T] = f32x4{0, 0, 0, 0}; }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    float s = 0.f;
    for (int T = 0; T < 16; ++T) s += acc[T][0] + acc[T][1] + acc[T][2] + acc[T][3] + prev[T][0];
    out[blockIdx.x * 512 + tid] = s;
}

// ---- one wavefront per SIMD (<= 512 VGPRs), two 16-sample groups per wavefront: each A fragment feeds six MFMAs ----
// ASMLDS: A fragments by untracked ds_read one batch ahead with manual lgkmcnt waits
template <bool ASMLDS, bool INTERLEAVE>
__global__ __launch_bounds__(256, 1) void k_probe2(const char* packed, const float* bias, float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < SLOTS * CHUNK / 4; i += 256) reinterpret_cast<float*>(smem)[i] = 1e-3f * (i & 255);
    float* s_bias = reinterpret_cast<float*>(smem + SLOTS * CHUNK);
    if (tid < 256) s_bias[tid] = bias[tid];
    __syncthreads();
    f32x4 acc[2][16], prev[2][16];
    for (int g = 0; g < 2; ++g)
        for (int T = 0; T < 16; ++T) { acc[g][T] = f32x4{0, 0, 0, 0}; prev[g][T] = f32x4{0.1f * lane + g, 0.2f, -0.3f, 0.4f * T}; }
    int slot = 0, islot = 3, ichunk = 3;
    for (int it0 = 0; it0 < iters; it0 += 8) {
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        half8 bh[2], bl[2];
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            float v[8];
            const float4 b0 = *reinterpret_cast<const float4*>(s_bias + 32 * s + 4 * (lane >> 4));
            const float4 b1 = *reinterpret_cast<const float4*>(s_bias + 32 * s + 16 + 4 * (lane >> 4));
            v[0] = fmaxf(prev[g][2 * s][0] + b0.x, 0.f); v[1] = fmaxf(prev[g][2 * s][1] + b0.y, 0.f);
            v[2] = fmaxf(prev[g][2 * s][2] + b0.z, 0.f); v[3] = fmaxf(prev[g][2 * s][3] + b0.w, 0.f);
            v[4] = fmaxf(prev[g][2 * s + 1][0] + b1.x, 0.f); v[5] = fmaxf(prev[g][2 * s + 1][1] + b1.y, 0.f);
            v[6] = fmaxf(prev[g][2 * s + 1][2] + b1.z, 0.f); v[7] = fmaxf(prev[g][2 * s + 1][3] + b1.w, 0.f);
            split8(v, bh[g], bl[g]);
        }
        auto handover = [&]() {
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const char* src = packed + (size_t)ichunk * CHUNK + wave * 8192 + lane * 16;
            char* dst = smem + islot * CHUNK + wave * 8192;
#pragma unroll
            for (int q = 0; q < 8; ++q)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + q * 1024),
                                                 (__attribute__((address_space(3))) void*)(dst + q * 1024), 16, 0, 0);
            ichunk = ichunk + 1 == NCH ? 0 : ichunk + 1;
            islot = islot + 1 == SLOTS ? 0 : islot + 1;
        };
        const char* base = smem + slot * CHUNK + lane * 16;
        slot = slot + 1 == SLOTS ? 0 : slot + 1;
        half8 ah[2][2], al[2][2];
        const unsigned laddr = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)base;
#define RD2(dst, piece) do { if (ASMLDS) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(laddr), "i"((piece) * 1024) : "memory"); else dst = *reinterpret_cast<const half8*>(base + (piece) * 1024); } while (0)
        RD2(ah[0][0], 0); RD2(al[0][0], 1); RD2(ah[0][1], 2); RD2(al[0][1], 3);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            if (b + 1 < 8) {
                RD2(ah[(b + 1) & 1][0], 4 * (b + 1)); RD2(al[(b + 1) & 1][0], 4 * (b + 1) + 1);
                RD2(ah[(b + 1) & 1][1], 4 * (b + 1) + 2); RD2(al[(b + 1) & 1][1], 4 * (b + 1) + 3);
            }
            if (b == 4) { if (ASMLDS) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); handover(); }
            if (ASMLDS) {
                if (b + 1 < 8 && b != 4)
                    asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(ah[b & 1][0]), "+v"(al[b & 1][0]), "+v"(ah[b & 1][1]), "+v"(al[b & 1][1]));
                else
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ah[b & 1][0]), "+v"(al[b & 1][0]), "+v"(ah[b & 1][1]), "+v"(al[b & 1][1]));
            }
            __builtin_amdgcn_sched_barrier(0);
            if (INTERLEAVE) {
                // four independent accumulators in rotation
                MF(acc[0][2 * b], ah[b & 1][0], bh[0]); MF(acc[1][2 * b], ah[b & 1][0], bh[1]);
                MF(acc[0][2 * b + 1], ah[b & 1][1], bh[0]); MF(acc[1][2 * b + 1], ah[b & 1][1], bh[1]);
                MF(acc[0][2 * b], ah[b & 1][0], bl[0]); MF(acc[1][2 * b], ah[b & 1][0], bl[1]);
                MF(acc[0][2 * b + 1], ah[b & 1][1], bl[0]); MF(acc[1][2 * b + 1], ah[b & 1][1], bl[1]);
                MF(acc[0][2 * b], al[b & 1][0], bh[0]); MF(acc[1][2 * b], al[b & 1][0], bh[1]);
                MF(acc[0][2 * b + 1], al[b & 1][1], bh[0]); MF(acc[1][2 * b + 1], al[b & 1][1], bh[1]);
            } else {
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int g = 0; g < 2; ++g) {
                        MF(acc[g][2 * b + t], ah[b & 1][t], bh[g]); MF(acc[g][2 * b + t], ah[b & 1][t], bl[g]); MF(acc[g][2 * b + t], al[b & 1][t], bh[g]);
                    }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int T = 0; T < 16; ++T) { prev[g][T] = acc[g][T]; acc[g][T] = f32x4{0, 0, 0, 0}; }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    float sres = 0.f;
    for (int g = 0; g < 2; ++g)
        for (int T = 0; T < 16; ++T) sres += acc[g][T][0] + acc[g][T][1] + acc[g][T][2] + acc[g][T][3] + prev[g][T][0];
    out[blockIdx.x * 256 + tid] = sres;
}

template <bool ASMLDS, bool INTERLEAVE>
void run2(const char* name, const char* packed, const float* bias, float* out, int iters) {
    const int lds = SLOTS * CHUNK + 1024;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_probe2<ASMLDS, INTERLEAVE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k_probe2<ASMLDS, INTERLEAVE>), dim3(256), dim3(256), lds, 0, packed, bias, out, iters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k_probe2<ASMLDS, INTERLEAVE>), dim3(256), dim3(256), lds, 0, packed, bias, out, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 3;
    const double flop = (double)iters * 256 * 4 * 96 * 16384.0;   // 4 wavefronts x 96 MFMAs per chunk: the same work
    printf("%-64s %8.3f ms  -> MLP16-equivalent %6.1f of 833\n", name, ms, flop / ms / 1e9 / 3);
}

template <int ORDER, int FENCE, bool EPI>
void run(const char* name, const char* packed, const float* bias, float* out, int iters) {
    const int lds = SLOTS * CHUNK + 1024;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_probe<ORDER, FENCE, EPI>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k_probe<ORDER, FENCE, EPI>), dim3(256), dim3(512), lds, 0, packed, bias, out, iters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k_probe<ORDER, FENCE, EPI>), dim3(256), dim3(512), lds, 0, packed, bias, out, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 3;
    const double flop = (double)iters * 256 * 8 * 48 * 16384.0;
    printf("%-64s %8.3f ms  -> MLP16-equivalent %6.1f of 833\n", name, ms, flop / ms / 1e9 / 3);
}

int main() {
    char* packed; float* out; float* bias;
    (void)hipMalloc(&packed, (size_t)NCH * CHUNK); { std::vector<_Float16> hbuf((size_t)NCH * CHUNK / 2); unsigned st = 12345u; for (auto& v : hbuf) { st = st * 1664525u + 1013904223u; v = (_Float16)(((int)(st >> 16) % 2001 - 1000) * 1e-3f); } (void)hipMemcpy(packed, hbuf.data(), hbuf.size() * 2, hipMemcpyHostToDevice); }
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&bias, 4096); { std::vector<float> bb(1024); for (int i = 0; i < 1024; ++i) bb[i] = 0.01f * (i % 37) - 0.1f; (void)hipMemcpy(bias, bb.data(), 4096, hipMemcpyHostToDevice); }
    const int iters = 72 * 20;
    run<0, 0, true>("dep x3 per tile, fenced batches (kernel today), epilogue", packed, bias, out, iters);
    run<1, 0, true>("two tiles interleaved, fenced batches, epilogue", packed, bias, out, iters);
    run<2, 0, true>("4 tiles hh/hl/lh, fenced batches, epilogue", packed, bias, out, iters);
    run<0, 1, true>("dep x3 per tile, no fences, epilogue", packed, bias, out, iters);
    run<1, 1, true>("two tiles interleaved, no fences, epilogue", packed, bias, out, iters);
    run<2, 1, true>("4 tiles hh/hl/lh, no fences, epilogue", packed, bias, out, iters);
    run<1, 2, true>("two tiles interleaved, sched_group_barrier mix, epilogue", packed, bias, out, iters);
    run<2, 2, true>("4 tiles hh/hl/lh, sched_group_barrier mix, epilogue", packed, bias, out, iters);
    run<4, 0, true>("asm in-place MFMA, dep x3 per tile, epilogue", packed, bias, out, iters);
    run<5, 0, true>("asm in-place MFMA, two tiles interleaved, epilogue", packed, bias, out, iters);
    run<3, 0, true>("asm ds_read + manual lgkmcnt, interleaved, fenced, epilogue", packed, bias, out, iters);
    run<3, 1, true>("asm ds_read + manual lgkmcnt, interleaved, no fences, epilogue", packed, bias, out, iters);
    run2<false, false>("1 wave/SIMD x 2 groups, tracked LDS, dep x3", packed, bias, out, iters);
    run2<false, true>("1 wave/SIMD x 2 groups, tracked LDS, 4 accumulators in rotation", packed, bias, out, iters);
    run2<true, false>("1 wave/SIMD x 2 groups, asm LDS double buffer, dep x3", packed, bias, out, iters);
    run2<true, true>("1 wave/SIMD x 2 groups, asm LDS double buffer, rotation", packed, bias, out, iters);
    run<0, 0, false>("dep x3 per tile, fenced, no epilogue", packed, bias, out, iters);
    run<2, 0, false>("4 tiles hh/hl/lh, fenced, no epilogue", packed, bias, out, iters);
    run<2, 1, false>("4 tiles hh/hl/lh, no fences, no epilogue", packed, bias, out, iters);
    return 0;
}
