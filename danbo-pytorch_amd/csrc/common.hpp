// Shared host/device helpers for the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include <atomic>
#include "../../include/danbo_hip.h"
#include "sample_math.hpp"

#define DANBO_CHECK_ARG(cond) do { if (!(cond)) return DANBO_EINVAL; } while (0)
#define DANBO_LAUNCH_RET() do { hipError_t e_ = hipGetLastError(); return e_ == hipSuccess ? 0 : (int)e_; } while (0)

// Opt a kernel into more than 64 KB of dynamic LDS.  The attribute is per DEVICE: the "done" mask is keyed by the calling
// thread's current device (a host that drives several GPUs from one process opts in on each), and it is an atomic so that
// concurrent host threads agree on it.  Returns the HIP error from the enclosing C-ABI function on failure.
#define DANBO_ENSURE_LDS(func, bytes)                                                                              \
    do {                                                                                                           \
        static std::atomic<unsigned long long> done_{0};                                                           \
        int dev_ = 0;                                                                                              \
        if (hipGetDevice(&dev_) != hipSuccess) return (int)hipGetLastError();                                      \
        const unsigned long long bit_ = 1ull << (dev_ & 63);                                                       \
        if (!(done_.load(std::memory_order_acquire) & bit_)) {                                                     \
            const hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(func),                         \
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes));   \
            if (e_ != hipSuccess) return (int)e_;                                                                  \
            done_.fetch_or(bit_, std::memory_order_release);                                                       \
        }                                                                                                          \
    } while (0)

// gfx950 erratum found in round 6 (tools/probe/cview_probe.hip, profiles/r06_pk_f32_erratum.txt): the packed fp32
// instructions v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 with op_sel = [x,1,..] -- the LOW half of the result computed from src1's HIGH
// dword -- return wrong results in lanes 48 .. 63 while ANOTHER wavefront on the same CU executes MFMA instructions (a few per 10^8
// beside MFMA alone, 2 per 10^4 beside MFMA + LDS reads); every other operand selection, and the scalar instructions, are exact.  It is
// what made the training step's view constants non-repeatable in round 4 (k_train_cview beside K2).  The compiler emits that form
// wherever it pairs two chains that share the odd register of an operand pair.  Kernels in which it did are compiled without packed
// fp32 instructions (device pass only: the host pass does not know the feature); tests/test_host_logic.py checks the ISA of the whole
// library for the form, so a kernel that acquires it fails the build's test instead of a training run.
#if defined(__HIP_DEVICE_COMPILE__)
#define DANBO_NO_PK_F32 __attribute__((target("no-packed-fp32-ops")))
#else
#define DANBO_NO_PK_F32
#endif

namespace danbo {

constexpr int WAVE = 64;
// Compute units of the calling thread's current device (hipDeviceAttributeMultiprocessorCount, cached per device): the
// persistent kernels size their grids from it.  256 on a full MI355X; partitioned (CPX / DPX) or harvested parts report less.
// Development switches (fault bisection, stream-order experiments, A/B of a packing) read the environment only in a
// -DDANBO_DEV_SWITCHES build (make DEV=1); the product build compiles every one of them to its default: no environment variable
// changes what the shipped library computes or in which order it enqueues it.
#ifdef DANBO_DEV_SWITCHES
static inline int dev_env(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
#else
static constexpr int dev_env(const char*, int dflt) { return dflt; }
#endif

static inline int num_cu() {
    static std::atomic<int> cached_[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    std::atomic<int>& c = cached_[dev & 63];
    int n = c.load(std::memory_order_relaxed);
    if (n == 0) {
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        { const int v = dev_env("DANBO_NUM_CU", 0); if (v > 0 && v <= n) n = v; }   // dev: size the grids for a CU-masked stream
        c.store(n, std::memory_order_relaxed);
    }
    return n;
}

static inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

// grid for a memory-bound grid-stride kernel: enough workgroups to fill 256 CUs x 8
static inline int stream_grid(long items, int block) {
    long g = (items + block - 1) / block;
    const long cap = (long)num_cu() * 8;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

// grid of a kernel whose workgroups loop over their share of the items: as many workgroups as are RESIDENT at once (what the
// kernel's registers / LDS allow per CU, asked from the runtime once per kernel), never more -- stream_grid's 8 per CU is two
// rounds for a kernel that fits 7 (k_composite_importance, 106 SGPRs: the eighth workgroup of every CU ran alone afterwards).
template <class K>
static inline int resident_grid(K kernel, long items, int block, size_t lds = 0) {
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, block, lds) != hipSuccess || per_cu < 1) per_cu = 1;
    long g = (items + block - 1) / block;
    const long cap = (long)num_cu() * per_cu;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// sin / cos of the positional encoding: Cody-Waite reduction by pi/2 (3 constants) + degree-7/8
// minimax polynomials; |error| < 2e-7 for |x| < 1e4, about a quarter of the code of OCML's sincosf
// (whose Payne-Hanek slow path is inlined at every call site).
__device__ __forceinline__ void pe_sincos(float x, float* s, float* c) {
    const float k = rintf(x * 0.636619772367581343f);  // 2/pi
    float r = fmaf(k, -1.57079601287841796875f, x);    // pi/2 split in three (24 + 24 + 24 bits)
    r = fmaf(k, -3.1391647326017846353352069854736328125e-7f, r);
    r = fmaf(k, -5.390302529957764765544681040410068817436695098876953125e-15f, r);
    const float r2 = r * r;
    float sp = fmaf(r2, 2.6083159809786593541502952575683593750e-6f, -1.981069071916863322258e-4f);
    sp = fmaf(sp, r2, 8.33307858556509017944e-3f);
    sp = fmaf(sp, r2, -1.66666597127914428711e-1f);
    const float sn = fmaf(sp * r2, r, r);
    float cp = fmaf(r2, 2.44331571593647822737693786621e-5f, -1.38873163610696792602539062500e-3f);
    cp = fmaf(cp, r2, 4.16666455566883087158203125e-2f);
    cp = fmaf(cp, r2, -0.5f);
    const float cs = fmaf(cp, r2, 1.0f);
    const int q = (int)k;
    const float s0 = (q & 1) ? cs : sn;
    const float c0 = (q & 1) ? sn : cs;
    *s = (q & 2) ? -s0 : s0;
    *c = ((q + 1) & 2) ? -c0 : c0;
}

// ---- wave64 scans on the DPP cross-lane paths of the VALU (no LDS crossbar traffic) ----
// inclusive scan: row_shr 1,2,4,8 inside each row of 16 lanes, then row_bcast15 / row_bcast31 carry the row
// totals across rows (gfx9 DPP controls; lanes without a source keep the identity)
#define DANBO_DPP_STEP(OP, IDENT, CTRL, ROWMASK)                                                              \
    {                                                                                                          \
        const int t_ = __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, (float)(IDENT)),                    \
                                                   __builtin_bit_cast(int, x), CTRL, ROWMASK, 0xf, false);     \
        x = OP(x, __builtin_bit_cast(float, t_));                                                              \
    }
__device__ __forceinline__ float dpp_add_(float a, float b) { return add_rn(a, b); }
__device__ __forceinline__ float dpp_mul_(float a, float b) { return mul_rn(a, b); }

__device__ __forceinline__ float wave_scan_add(float x) {
    DANBO_DPP_STEP(dpp_add_, 0.f, 0x111, 0xf) DANBO_DPP_STEP(dpp_add_, 0.f, 0x112, 0xf)
    DANBO_DPP_STEP(dpp_add_, 0.f, 0x114, 0xf) DANBO_DPP_STEP(dpp_add_, 0.f, 0x118, 0xf)
    DANBO_DPP_STEP(dpp_add_, 0.f, 0x142, 0xa) DANBO_DPP_STEP(dpp_add_, 0.f, 0x143, 0xc)
    return x;
}
__device__ __forceinline__ float wave_scan_mul(float x) {
    DANBO_DPP_STEP(dpp_mul_, 1.f, 0x111, 0xf) DANBO_DPP_STEP(dpp_mul_, 1.f, 0x112, 0xf)
    DANBO_DPP_STEP(dpp_mul_, 1.f, 0x114, 0xf) DANBO_DPP_STEP(dpp_mul_, 1.f, 0x118, 0xf)
    DANBO_DPP_STEP(dpp_mul_, 1.f, 0x142, 0xa) DANBO_DPP_STEP(dpp_mul_, 1.f, 0x143, 0xc)
    return x;
}
// bitwise OR over the wave, in every lane (same DPP tree)
__device__ __forceinline__ uint32_t wave_or(uint32_t x) {
#define DANBO_DPP_OR(CTRL, ROWMASK) x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, ROWMASK, 0xf, false);
    DANBO_DPP_OR(0x111, 0xf) DANBO_DPP_OR(0x112, 0xf) DANBO_DPP_OR(0x114, 0xf) DANBO_DPP_OR(0x118, 0xf)
    DANBO_DPP_OR(0x142, 0xa) DANBO_DPP_OR(0x143, 0xc)
#undef DANBO_DPP_OR
    return (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
}
// total of the wave in every lane (summation order = the scan's tree)
__device__ __forceinline__ float wave_total(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wave_scan_add(x)), 63));
}

// x of lane ^ 32 / lane ^ 16 on the VALU (gfx950 v_permlane32_swap / v_permlane16_swap + one select) instead of the
// LDS-crossbar ds_bpermute that __shfl_xor compiles to
__device__ __forceinline__ float lane_xor32(float x) {
    const unsigned u = __builtin_bit_cast(unsigned, x);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);   // r[0]: upper half = my lower half; r[1]: lower = my upper
    return __builtin_bit_cast(float, (threadIdx.x & 32) ? r[0] : r[1]);
}
__device__ __forceinline__ float lane_xor16(float x) {
    const unsigned u = __builtin_bit_cast(unsigned, x);
    const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);   // r[0]: odd rows = my even rows; r[1]: even = my odd
    return __builtin_bit_cast(float, (threadIdx.x & 16) ? r[0] : r[1]);
}

// Zeroing / small copies as KERNELS, not hipMemsetAsync / hipMemcpyAsync: inside a captured HIP graph (torch.cuda.graph around the
// training step) a byte-granular memset node of an odd-sized region was replayed wrongly on this ROCm (the second replay left
// the step's device counters non-zero and the next kernel wrote out of bounds: "Memory access fault"); eager launches were fine.
// Kernels replay like every other node.  One copy per translation unit (static).
static __global__ __launch_bounds__(256) void k_zero_words_(uint32_t* __restrict__ p0, long n0, uint32_t* __restrict__ p1, long n1) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n0; i += stride) p0[i] = 0u;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n1; i += stride) p1[i] = 0u;
}
static inline void zero_words(void* p0, long n0, void* p1, long n1, hipStream_t st) {
    const long n = n0 > n1 ? n0 : n1;
    hipLaunchKernelGGL(k_zero_words_, dim3(stream_grid(n, 256)), dim3(256), 0, st, static_cast<uint32_t*>(p0), n0,
                       static_cast<uint32_t*>(p1), p1 ? n1 : 0);
}
static __global__ __launch_bounds__(64) void k_copy_words_(const uint32_t* __restrict__ s0, uint32_t* __restrict__ d0, int n0,
                                                           const uint32_t* __restrict__ s1, uint32_t* __restrict__ d1, int n1) {
    for (int i = threadIdx.x; i < n0; i += 64) d0[i] = s0[i];
    for (int i = threadIdx.x; i < n1; i += 64) d1[i] = s1[i];
}

// number of rows a kernel has to process: device-side count (clamped to capacity) or host n
__device__ __forceinline__ int resolve_count(const int32_t* count, int n_cap) {
    if (count == nullptr) return n_cap;
    int c = *count;
    return c < n_cap ? c : n_cap;
}

// hi = fp16(x), lo = fp16(x - float(hi)) of eight values as THREE instructions per pair: v_cvt_pk_f16_f32 for the two hi halves,
// v_fma_mixlo_f16 / v_fma_mixhi_f16 (fma(x, 1.0, -hi) with the fp16 operand widened inside the instruction, the fp32 result -- exact:
// x - hi has at most 13 significant bits -- rounded once to fp16 into the low / high half of the destination) for the two lo halves.
// What the compiler makes of the plain C++ form is ~7 per pair (cvt, cvt back, sub -- partly as v_pk_add_f32, an anti-lever beside
// MFMAs -- and a second cvt_pk): 43 VALU instructions per k-step epilogue of K3, 28 with this.  Bit-identical on 33.5 M random pairs of
// every exponent incl. the fp16 subnormal and overflow ranges (tools/probe/split_probe.hip).
// HAZARD: these are inline-asm statements, which the compiler's hazard recognizer does not see as VALU writes -- an MFMA *builtin*
// that reads hi / lo within a couple of instructions of the split gets stale registers (K2, round 5: h wrong by 1e-2 when its
// layer-0 MFMAs moved right behind the split).  A consumer that close writes the wait states out (k_assign16.hip a16_split8;
// the asm MFMA chunks of mlp16_core.hpp start with `s_nop 1` for the same reason).
template <class H8>
__device__ __forceinline__ void split8_mix(const float* v, H8& hi, H8& lo) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 h, l;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        unsigned hp, lp;
        asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hp) : "v"(v[2 * p]), "v"(v[2 * p + 1]));
        asm("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(lp) : "v"(v[2 * p]), "v"(hp));
        asm("v_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(lp) : "v"(v[2 * p + 1]), "v"(hp));
        h[p] = hp;
        l[p] = lp;
    }
    hi = __builtin_bit_cast(H8, h);
    lo = __builtin_bit_cast(H8, l);
}


}  // namespace danbo
