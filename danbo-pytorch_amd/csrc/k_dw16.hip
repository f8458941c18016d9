// Weight / bias gradients of the dense layers of the training step, all layers in ONE launch:
//     dW_l [N, K] = dz_l^T [N x rows] * x_l [rows x K],      db_l [N] = sum_rows dz_l
// (reference: what loss.backward() computes for every nn.Linear of core/networks/nerf.py:176-209, core/trainer.py:563-576),
// with fp32-accurate products on the fp16 matrix cores (hi/lo split as in k_linear16.hip, fp32 accumulate).  gfx950 only.
//
// The reduction index of this GEMM is the ROW of both operands, i.e. the slow index of both row-major buffers, while an MFMA
// fragment wants 8 consecutive reduction indices per lane.  So a workgroup stages 32 rows of its 256 gradient columns and
// 256 input columns through LDS *transposed* (a whole 256 x 256 layer per workgroup: every operand row is read from HBM ONCE
// per layer -- with 128-column tiles the kernel re-read the inputs per tile and sat on the HBM roof at 3.4 TB/s): every thread loads the same 4 columns of two consecutive rows (2 x 16 B),
// splits them into fp16 hi / lo and writes row PAIRS as 32-bit words at [column][row] (column stride 80 B: 2-way bank
// conflicts at most for the writes and for the 16-byte fragment reads).  Eight wavefronts then own 64 x 128 of the
// 256 x 256 output tile each (32 accumulator tiles = 128 VGPRs): per 32-row step 24 fragment reads feed 96 MFMAs.
// Rows are split over `slices` workgroups per tile; partial tiles go to a scratch buffer and a second kernel sums the
// slices in a fixed order (deterministic, no float atomics), undoes the power-of-two pre-scale of the gradient operand
// (in_maxabs of danbo_linear16_ex) and writes the gradients in nn.Linear layout.
// Algorithmic traffic per row and layer: 4 (N + K) bytes read once per 128 x 256 tile of the layer; flops 2 N K (x 3 MFMA products).
#include "common.hpp"

namespace danbo {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int DW_TN = 256, DW_TK = 256, DW_ROWS = 32, DW_THREADS = 512;
constexpr int DW_CSTRIDE = 80;                           // bytes per column in LDS: 32 rows x 2 B + 16 B padding
constexpr int DW_A_BYTES = DW_TN * DW_CSTRIDE;           // one of hi / lo
constexpr int DW_B_BYTES = DW_TK * DW_CSTRIDE;
constexpr int DW_LDS_BYTES = 2 * DW_A_BYTES + 2 * DW_B_BYTES + DW_TN * 4;   // + column sums
constexpr int DW_MAX_LAYERS = 12;

struct DwLayer {
    const float* dy;       // [rows, ldy] gradient with respect to the layer's pre-activation
    const float* x1;       // [rows, ld1]
    const float* x2;       // [rows, ld2] or nullptr
    const float* dy_maxabs;  // device scalar: max |dy| (power-of-two pre-scale), or nullptr
    float* gw;             // nn.Linear.weight.grad [N, K1 + K2] rows < split_n ...
    float* gw2;            // ... and rows >= split_n (feature_linear | alpha_linear evaluated as one layer), or nullptr
    float* gb;
    float* gb2;
    int ldy, ld1, ld2, N, K1, K2, split_n;
    int K1p, Kv;           // K1 rounded up to 4; virtual width K1p + K2
    int tile0, nt_k;       // first tile of this layer, tiles along K
    long part_off;         // floats: this layer's [slices][N * Kv + N] partial block
};

struct DwArgs {
    DwLayer l[DW_MAX_LAYERS];
    int n_layers, n_tiles, slices, M;
    const int32_t* count;  // device row count (or nullptr: M)
    float* part;
};

__device__ __forceinline__ int dw_rows_per_slice(int M, int slices) {
    const int r = (M + slices - 1) / slices;
    return (r + DW_ROWS - 1) / DW_ROWS * DW_ROWS;
}

__device__ __forceinline__ void dw_pow2_scale(float maxabs, float& s, float& inv) {
    const unsigned E = (__builtin_bit_cast(unsigned, maxabs) >> 23) & 255u;
    unsigned se = (E == 0u || E == 255u) ? 127u : 257u - E;    // max * s in [8, 16)
    se = se < 1u ? 1u : (se > 253u ? 253u : se);
    s = __builtin_bit_cast(float, se << 23);
    inv = __builtin_bit_cast(float, (254u - se) << 23);
}

// two rows x 4 columns -> hi / lo halves, stored as row pairs
__device__ __forceinline__ void dw_store4(char* hi_base, char* lo_base, int col, int rp, const f32x4& r0, const f32x4& r1, float sc) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float a = r0[j] * sc, b = r1[j] * sc;
        const _Float16 ah = (_Float16)a, bh = (_Float16)b;
        const half2v h = {ah, bh};
        const half2v l = {(_Float16)(a - (float)ah), (_Float16)(b - (float)bh)};
        *reinterpret_cast<half2v*>(hi_base + (col + j) * DW_CSTRIDE + rp * 4) = h;
        *reinterpret_cast<half2v*>(lo_base + (col + j) * DW_CSTRIDE + rp * 4) = l;
    }
}

__global__ __launch_bounds__(DW_THREADS, 1) void k_dw16(DwArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* s_ah = smem;
    char* s_al = smem + DW_A_BYTES;
    char* s_bh = smem + 2 * DW_A_BYTES;
    char* s_bl = smem + 2 * DW_A_BYTES + DW_B_BYTES;
    float* s_db = reinterpret_cast<float*>(smem + 2 * DW_A_BYTES + 2 * DW_B_BYTES);

    const int M = resolve_count(a.count, a.M);
    const int sl = blockIdx.y;
    const int rps = dw_rows_per_slice(M, a.slices);
    const int row0 = sl * rps;
    if (row0 >= M) return;                       // empty slice: the reduction only reads slices that exist
    const int row_end = min(row0 + rps, M);

    // which layer / tile
    int li = 0;
    while (li + 1 < a.n_layers && (int)blockIdx.x >= a.l[li + 1].tile0) ++li;
    const DwLayer& L = a.l[li];
    const int t = (int)blockIdx.x - L.tile0;
    const int tn = t / L.nt_k, tk = t % L.nt_k;
    const int n0 = tn * DW_TN, v0 = tk * DW_TK;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float sc = 1.f, inv = 1.f;
    if (L.dy_maxabs != nullptr) dw_pow2_scale(*L.dy_maxabs, sc, inv);

    // loader mapping: lanes run over row pairs first (16 pairs = 32 rows), then over groups of 4 columns
    const int rp = tid & 15;
    const int ga = tid >> 4;                     // columns 4 ga .. + 3 and 4 (ga + 32) .. + 3 of both operands
    f32x4 ra[2][2], rb[2][2];                    // [group][row of the pair]
    // 32-bit byte offsets from the (uniform) buffer bases: 12 loads in flight would otherwise pin 24 address registers
    const char* const dy_b = reinterpret_cast<const char*>(L.dy);
    const char* const x1_b = reinterpret_cast<const char*>(L.x1);
    const char* const x2_b = reinterpret_cast<const char*>(L.x2);

    auto load_step = [&](int r) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int col = n0 + 4 * (ga + 32 * u);
#pragma unroll
            for (int w = 0; w < 2; ++w) {
                const int row = r + 2 * rp + w;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (row < row_end && col < L.N) {
                    const unsigned off = ((unsigned)row * (unsigned)L.ldy + (unsigned)col) * 4u;
                    const float* p = reinterpret_cast<const float*>(dy_b + off);
                    if (col + 4 <= L.N) v = *reinterpret_cast<const f32x4*>(p);
                    else {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (col + j < L.N) v[j] = p[j];
                    }
                }
                ra[u][w] = v;
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int vc = v0 + 4 * (ga + 32 * u);       // virtual column: [x1 padded to K1p | x2]
#pragma unroll
            for (int w = 0; w < 2; ++w) {
                const int row = r + 2 * rp + w;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (row < row_end && vc < L.Kv) {
                    if (vc < L.K1p) v = *reinterpret_cast<const f32x4*>(x1_b + ((unsigned)row * (unsigned)L.ld1 + (unsigned)vc) * 4u);
                    else {
                        const int c2 = vc - L.K1p;
                        const float* p = reinterpret_cast<const float*>(x2_b + ((unsigned)row * (unsigned)L.ld2 + (unsigned)c2) * 4u);
                        if (c2 + 4 <= L.K2) v = *reinterpret_cast<const f32x4*>(p);
                        else {
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                if (c2 + j < L.K2) v[j] = p[j];
                        }
                    }
                }
                rb[u][w] = v;
            }
        }
    };

    // wave tile: 64 gradient columns x 128 input columns (8 wavefronts: 4 x 2)
    const int wn = (wave >> 1) * 64, wk = (wave & 1) * 128;
    const int m = lane & 15, q = lane >> 4;
    const bool wave_live = (n0 + wn < L.N) && (v0 + wk < L.Kv);
    f32x4 acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int i = tid; i < DW_TN; i += DW_THREADS) s_db[i] = 0.f;
    load_step(row0);
    __syncthreads();
    for (int r = row0; r < row_end; r += DW_ROWS) {
        // registers -> LDS (transposed, split)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            dw_store4(s_ah, s_al, 4 * (ga + 32 * u), rp, ra[u][0], ra[u][1], sc);
            if (tk == 0) {   // bias gradient: the 16 lanes of a DPP row hold the 32 rows of the same 4 columns
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float x = ra[u][0][j] + ra[u][1][j];
                    DANBO_DPP_STEP(dpp_add_, 0.f, 0x111, 0xf) DANBO_DPP_STEP(dpp_add_, 0.f, 0x112, 0xf)
                    DANBO_DPP_STEP(dpp_add_, 0.f, 0x114, 0xf) DANBO_DPP_STEP(dpp_add_, 0.f, 0x118, 0xf)
                    if (rp == 15) s_db[4 * (ga + 32 * u) + j] += x;      // this lane is the only writer of the entry
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) dw_store4(s_bh, s_bl, 4 * (ga + 32 * u), rp, rb[u][0], rb[u][1], 1.f);
        __syncthreads();
        if (r + DW_ROWS < row_end) load_step(r + DW_ROWS);     // next step's rows fly under this step's MFMAs
        if (wave_live) {
            half8 ah[4], al[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ah[i] = *reinterpret_cast<const half8*>(s_ah + (wn + 16 * i + m) * DW_CSTRIDE + q * 16);
                al[i] = *reinterpret_cast<const half8*>(s_al + (wn + 16 * i + m) * DW_CSTRIDE + q * 16);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const half8 bh = *reinterpret_cast<const half8*>(s_bh + (wk + 16 * j + m) * DW_CSTRIDE + q * 16);
                const half8 bl = *reinterpret_cast<const half8*>(s_bl + (wk + 16 * j + m) * DW_CSTRIDE + q * 16);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bh, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bl, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[i], bh, acc[i][j], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }

    // partial tile: lane (m, q) of accumulator tile (i, j) holds dW[n0 + wn + 16 i + 4 q + e][v0 + wk + 16 j + m]
    float* part = a.part + L.part_off + (long)sl * ((long)L.N * L.Kv + L.N);
    if (wave_live) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int vc = v0 + wk + 16 * j + m;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int n = n0 + wn + 16 * i + 4 * q + e;
                    if (n < L.N && vc < L.Kv) part[(long)n * L.Kv + vc] = acc[i][j][e];
                }
            }
    }
    // column sums of the gradient (bias gradient), once per gradient-column tile
    if (tk == 0) {
        __syncthreads();
        for (int i = tid; i < DW_TN; i += DW_THREADS)
            if (n0 + i < L.N) part[(long)L.N * L.Kv + n0 + i] = s_db[i];
    }
}

// sums the slices in order, scales back, writes nn.Linear layout
__global__ __launch_bounds__(256) void k_dw16_reduce(DwArgs a) {
    const int M = resolve_count(a.count, a.M);
    const int rps = dw_rows_per_slice(M, a.slices);
    const int live = M > 0 ? (M + rps - 1) / rps : 0;
    const DwLayer& L = a.l[blockIdx.y];
    float sc = 1.f, inv = 1.f;
    if (L.dy_maxabs != nullptr) dw_pow2_scale(*L.dy_maxabs, sc, inv);
    const int K = L.K1 + L.K2;
    const long per = (long)L.N * L.Kv + L.N;
    const long total = (long)L.N * K + L.N;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const bool is_b = idx >= (long)L.N * K;
        int n, k = 0;
        long src;
        if (is_b) {
            n = (int)(idx - (long)L.N * K);
            src = (long)L.N * L.Kv + n;
        } else {
            n = (int)(idx / K);
            k = (int)(idx % K);
            src = (long)n * L.Kv + (k < L.K1 ? k : k - L.K1 + L.K1p);
        }
        float s = 0.f;
        const float* p = a.part + L.part_off + src;
        int i = 0;
        for (; i + 4 <= live; i += 4) {          // four slices in flight (same summation order as the plain loop)
            const float v0 = p[(long)i * per], v1 = p[(long)(i + 1) * per], v2 = p[(long)(i + 2) * per], v3 = p[(long)(i + 3) * per];
            s += v0; s += v1; s += v2; s += v3;
        }
        for (; i < live; ++i) s += p[(long)i * per];
        if (is_b) {
            float* g = n < L.split_n ? L.gb : L.gb2;
            if (g != nullptr) g[n < L.split_n ? n : n - L.split_n] = s;   // the bias sums use the unscaled gradient
        } else {
            float* g = n < L.split_n ? L.gw : L.gw2;
            if (g != nullptr) g[(long)(n < L.split_n ? n : n - L.split_n) * K + k] = s * inv;
        }
    }
}

}  // namespace danbo

using namespace danbo;

extern "C" long danbo_dw16_scratch_floats(const DanboDwLayer* layers, int n_layers, int slices) {
    if (!layers || n_layers < 1 || n_layers > DW_MAX_LAYERS || slices < 1) return -1;
    long total = 0;
    for (int i = 0; i < n_layers; ++i) {
        const long K1p = (layers[i].K1 + 3) & ~3;
        total += (long)slices * ((long)layers[i].N * (K1p + layers[i].K2) + layers[i].N);
    }
    return total;
}

extern "C" int danbo_dw16(const DanboDwLayer* layers, int n_layers, int M, const int32_t* count, int slices, float* scratch,
                          void* stream) {
    DANBO_CHECK_ARG(layers && n_layers >= 1 && n_layers <= DW_MAX_LAYERS && M >= 0 && slices >= 1 && slices <= 1024 && scratch);
    if (M == 0) return 0;
    DwArgs a;
    a.n_layers = n_layers;
    a.slices = slices;
    a.M = M;
    a.count = count;
    a.part = scratch;
    int tile = 0;
    long off = 0;
    for (int i = 0; i < n_layers; ++i) {
        const DanboDwLayer& s = layers[i];
        DANBO_CHECK_ARG(s.dy && s.x1 && s.N >= 1 && s.K1 >= 1 && s.K2 >= 0 && (s.K2 == 0 || s.x2));
        DANBO_CHECK_ARG(s.ld1 % 4 == 0 && s.ld1 >= ((s.K1 + 3) & ~3) && (uintptr_t)s.x1 % 16 == 0);
        DANBO_CHECK_ARG(s.K2 == 0 || (s.ld2 % 4 == 0 && s.ld2 >= s.K2 && (uintptr_t)s.x2 % 16 == 0));
        DANBO_CHECK_ARG(s.ldy % 4 == 0 && s.ldy >= s.N && (uintptr_t)s.dy % 16 == 0);
        DwLayer& L = a.l[i];
        L.dy = s.dy; L.x1 = s.x1; L.x2 = s.x2; L.dy_maxabs = s.dy_maxabs;
        L.gw = s.gw; L.gw2 = s.gw2; L.gb = s.gb; L.gb2 = s.gb2;
        L.ldy = s.ldy; L.ld1 = s.ld1; L.ld2 = s.ld2; L.N = s.N; L.K1 = s.K1; L.K2 = s.K2;
        L.split_n = s.gw2 || s.gb2 ? s.split_n : 0x7fffffff;
        L.K1p = (s.K1 + 3) & ~3;
        L.Kv = L.K1p + s.K2;
        L.tile0 = tile;
        L.nt_k = (L.Kv + DW_TK - 1) / DW_TK;
        tile += ((s.N + DW_TN - 1) / DW_TN) * L.nt_k;
        L.part_off = off;
        off += (long)slices * ((long)s.N * L.Kv + s.N);
    }
    a.n_tiles = tile;
    DANBO_ENSURE_LDS(k_dw16, DW_LDS_BYTES);
    hipLaunchKernelGGL(k_dw16, dim3(tile, slices), dim3(DW_THREADS), DW_LDS_BYTES, (hipStream_t)stream, a);
    hipLaunchKernelGGL(k_dw16_reduce, dim3(128, n_layers), dim3(256), 0, (hipStream_t)stream, a);
    DANBO_LAUNCH_RET();
}
