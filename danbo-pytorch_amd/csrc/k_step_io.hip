// What a training step needs in front of its first kernel, in two launches instead of ten.
//
// (1) The random draws of a training step in ONE launch: stratified offsets + inverse-CDF uniforms (sample_from_lineseg /
// isample_from_lineseg, reference core/utils/ray_utils.py:206-291: torch.rand) and the density noise of both passes
// (NeRF.raw2outputs, core/networks/nerf.py:316: torch.randn * raw_noise_std) from a counter-based generator (Philox4x32-10)
// whose state lives on the DEVICE: the kernel advances it itself, so a captured HIP graph draws fresh numbers on every replay
// without host-side bookkeeping.  (torch's own generators inside a captured graph cost two fill launches in front of every
// replay plus one launch per distribution: four ~5 us nodes in front of the step's first kernel instead of one.)
//
// (2) The batch's tensors gathered into the captured graph's static input buffer in ONE launch: rows of any stride (the loader
// hands over per-RAY copies of the per-pose tensors -- skts [R,24,4,4], bones, cyls -- of which every (R / G)-th row is taken), so
// the strided slices need no .contiguous() launch each in front of a torch.cat.  gfx950 only.
#include "common.hpp"
#include "../../include/danbo_hip.h"

namespace danbo {

__device__ __forceinline__ uint4 philox4x32_10(uint4 c, uint2 k) {
#pragma unroll
    for (int round = 0; round < 10; ++round) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c.x), lo0 = 0xD2511F53u * c.x;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c.z), lo1 = 0xCD9E8D57u * c.z;
        c = uint4{hi1 ^ c.y ^ k.x, lo1, hi0 ^ c.w ^ k.y, lo0};
        k.x += 0x9E3779B9u;
        k.y += 0xBB67AE85u;
    }
    return c;
}

// state: [0] seed, [1] offset (Philox counter of the next call's thread 0), [2] ticket of the running launch (0 between launches).
// Thread i of the launch owns counter offset + i: words 4i..4i+3 of the uniform stream (counter word 2 = 0) and of the normal
// stream (counter word 2 = 1; two Box-Muller pairs).  The LAST workgroup to finish advances the offset: every workgroup has
// read it by then.
__global__ __launch_bounds__(256) void k_random_draws(unsigned long long* __restrict__ state, long n_uniform,
                                                      float* __restrict__ uniform, long n_normal, float normal_std,
                                                      float* __restrict__ normal) {
    __shared__ unsigned long long s_off;
    if (threadIdx.x == 0) s_off = __hip_atomic_load(state + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const unsigned long long seed = state[0];
    const uint2 key{(uint32_t)seed, (uint32_t)(seed >> 32)};
    const long quads = (max(n_uniform, n_normal) + 3) / 4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < quads; i += (long)gridDim.x * blockDim.x) {
        const unsigned long long ctr = s_off + (unsigned long long)i;
        if (4 * i < n_uniform) {
            const uint4 x = philox4x32_10(uint4{(uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u}, key);
            const uint32_t w[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (4 * i + e < n_uniform) uniform[4 * i + e] = (float)(w[e] >> 8) * 0x1p-24f;          // [0, 1), 24 bits
        }
        if (4 * i < n_normal) {
            const uint4 x = philox4x32_10(uint4{(uint32_t)ctr, (uint32_t)(ctr >> 32), 1u, 0u}, key);
            const uint32_t w[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
            for (int e = 0; e < 4; e += 2) {
                // Box-Muller on the hardware's log2 / sin / cos (v_sin_f32 and v_cos_f32 take REVOLUTIONS: no 2 pi, no argument
                // reduction for [0, 1)); their ~1e-6 absolute error is far below what a noise draw needs
                const float u1 = (float)((w[e] >> 8) + 1u) * 0x1p-24f;                               // (0, 1]
                const float rad = __builtin_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1)) * normal_std;   // -2 ln u1
                const float rev = (float)(w[e + 1] >> 8) * 0x1p-24f;
                const float sn = __builtin_amdgcn_sinf(rev), cs = __builtin_amdgcn_cosf(rev);
                if (4 * i + e < n_normal) normal[4 * i + e] = rad * cs;
                if (4 * i + e + 1 < n_normal) normal[4 * i + e + 1] = rad * sn;
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        const unsigned long long done = atomicAdd(state + 2, 1ull);
        if (done + 1ull == (unsigned long long)gridDim.x) {
            state[1] = s_off + (unsigned long long)quads;
            state[2] = 0ull;
        }
    }
}

struct GatherArgs {
    DanboRowSpan span[DANBO_MAX_ROW_SPANS];
    long first[DANBO_MAX_ROW_SPANS + 1];      // first 32-bit word (of the launch's flat index space) of span i; [n] = total
    int n;
};

__global__ __launch_bounds__(256) void k_gather_rows(GatherArgs a, uint32_t* __restrict__ dst) {
    const long total = a.first[a.n];
    for (long w = (long)blockIdx.x * blockDim.x + threadIdx.x; w < total; w += (long)gridDim.x * blockDim.x) {
        int i = 0;
#pragma unroll
        for (int k = 1; k < DANBO_MAX_ROW_SPANS; ++k) i += (k < a.n && w >= a.first[k]) ? 1 : 0;
        const DanboRowSpan& sp = a.span[i];
        const long v = w - a.first[i];
        const long row = v / sp.row_words, col = v - row * sp.row_words;
        dst[sp.dst_word + v] = static_cast<const uint32_t*>(sp.src)[row * sp.src_row_stride_words + col];
    }
}



// Parameter-sized products of a weight refresh (a folded head matrix, a per-camera table, a bias pushed through a layer): one
// thread per output, the sum over k in ascending order accumulated in float64 and rounded once -- what the hosts computed with
// float64 torch matmuls (library GEMMs) until round 5.  Not for per-sample work.
__global__ __launch_bounds__(256) void k_small_matmul(const float* __restrict__ A, long sa_m, long sa_k, const float* __restrict__ Bm, long sb_k,
                                                      long sb_n, const float* __restrict__ bias, int M, int N, int K, float* __restrict__ C,
                                                      long ldc) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (long)M * N; i += (long)gridDim.x * blockDim.x) {
        const int m = (int)(i / N), n = (int)(i % N);
        double acc = bias ? (double)bias[n] : 0.0;
        const float* a = A + m * sa_m;
        const float* b = Bm + n * sb_n;
        for (int k = 0; k < K; ++k) acc += (double)a[k * sa_k] * (double)b[k * sb_k];
        C[m * ldc + n] = (float)acc;
    }
}

}  // namespace danbo

using namespace danbo;

extern "C" int danbo_small_matmul(const float* A, long sa_m, long sa_k, const float* B, long sb_k, long sb_n, const float* bias, int M, int N,
                                  int K, float* C, long ldc, void* stream) {
    DANBO_CHECK_ARG(A && B && C && M >= 1 && N >= 1 && K >= 1 && ldc >= N);
    hipLaunchKernelGGL(k_small_matmul, dim3(stream_grid((long)M * N, 256)), dim3(256), 0, (hipStream_t)stream, A, sa_m, sa_k, B, sb_k, sb_n, bias,
                       M, N, K, C, ldc);
    DANBO_LAUNCH_RET();
}

extern "C" int danbo_gather_rows(const DanboRowSpan* spans, int n_spans, void* dst, void* stream) {
    DANBO_CHECK_ARG(spans != nullptr && dst != nullptr && n_spans >= 1 && n_spans <= DANBO_MAX_ROW_SPANS);
    GatherArgs a;
    long total = 0;
    for (int i = 0; i < n_spans; ++i) {
        DANBO_CHECK_ARG(spans[i].src != nullptr && spans[i].rows >= 1 && spans[i].row_words >= 1 && spans[i].dst_word >= 0 &&
                        spans[i].src_row_stride_words >= 0);
        a.span[i] = spans[i];
        a.first[i] = total;
        total += (long)spans[i].rows * spans[i].row_words;
    }
    for (int i = n_spans; i < DANBO_MAX_ROW_SPANS; ++i) { a.span[i] = spans[0]; a.first[i] = total; }
    a.first[DANBO_MAX_ROW_SPANS] = total;
    a.first[n_spans] = total;
    a.n = n_spans;
    hipLaunchKernelGGL(k_gather_rows, dim3(stream_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, a, static_cast<uint32_t*>(dst));
    DANBO_LAUNCH_RET();
}

extern "C" int danbo_random_draws(uint64_t* state, long n_uniform, float* uniform, long n_normal, float normal_std, float* normal,
                                  void* stream) {
    DANBO_CHECK_ARG(state != nullptr && n_uniform >= 0 && n_normal >= 0 && (n_uniform == 0 || uniform) && (n_normal == 0 || normal));
    DANBO_CHECK_ARG(n_uniform + n_normal > 0 && normal_std >= 0.f);
    const long quads = (std::max(n_uniform, n_normal) + 3) / 4;
    hipLaunchKernelGGL(k_random_draws, dim3(stream_grid(quads, 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<unsigned long long*>(state), n_uniform, uniform, n_normal, normal_std, normal);
    DANBO_LAUNCH_RET();
}
