// K3 (fast variant): the DANBO density/colour MLP with fp32-accurate products on the
// half-precision matrix cores.  gfx950 only.
//
// Numerics: every fp32 operand x is split as x = hi + lo with hi = fp16(x), lo = fp16(x - hi)
// (22 significant bits together) and a product is evaluated as hi*hi + hi*lo + lo*hi on
// v_mfma_f32_16x16x32_f16 with fp32 accumulation -- the dropped lo*lo term is 2^-22 relative.
// Measured against fp64 on this network: max 2e-6 relative on the raw logits, the same class as
// a plain fp32 GEMM (1e-6) and 50x inside the 1e-4 bound of the north-star.  Three half
// MFMAs replace sixteen-times-slower fp32 MFMAs: 5.3x the matrix rate of k_pe_mlp.
//
// Dataflow (transposed chain, activations never leave registers):
//   out^T [features x samples] = W [features x k] * act^T [k x samples]
//   A operand = weights, shared by every wavefront: streamed once per workgroup through a 4-slot
//               32 KB LDS ring filled by global_load_lds, pre-packed in fragment order;
//   B operand = this wavefront's 16 samples: lane = sample m + 16*q, k-slots 8q..8q+7;
//   D tile T  : lane (m, q) holds features 16T + 4q + i (i = reg 0..3) of sample m -- exactly the
//   B-fragment layout of the next layer once its k-step s is defined to cover tiles 2s and 2s+1
//   (the pack kernel permutes the weight columns accordingly), so bias + ReLU + hi/lo split stay
//   in registers and feed the next GEMM.
// One workgroup = 8 wavefronts x 16 samples = 128 rows, TWO wavefronts per SIMD (<= 256 VGPRs
// each): while one wavefront waits on the weight ring, LDS or its epilogue VALU work, the other
// keeps the matrix pipe busy (a one-wavefront-per-SIMD version spent half its time stalled).
#include <cstdlib>
#include "mlp16_core.hpp"

namespace danbo {

constexpr int NCH_X0 = 7;              // 7 k-steps of 32 PE features (208 >= 195)
constexpr int NCH_ACT = 8;             // 8 k-steps of 32 features
constexpr int NCH_VIEW = 4;            // 8 k-steps, 8 output tiles -> 2 k-steps per chunk
// feature_linear (256->256, no activation) and the per-sample part of views_linears.0 (256->128) are
// merged into ONE 256->128 GEMM at pack time: W_fv = W_v[:, :256] W_f, bias W_v[:, :256] b_f folded into
// the per-ray view constants (65 536 fewer MACs per row than the reference's 677 376)
constexpr int NCH_TOTAL = NCH_X0 + 4 * NCH_ACT + (NCH_X0 + NCH_ACT) + 2 * NCH_ACT + NCH_VIEW;  // 74
static_assert(NCH_TOTAL * CHUNK_BYTES + DANBO_MLP16_TRAILER_BYTES == DANBO_MLP16_PACKED_BYTES, "header constant out of date");
constexpr int X0_KSTEPS = 7;
constexpr int IN_CH = 195, W_ = 256, VW_ = 128;

// ---------------------------------------------------------------------------------------------
// weight packing, two launches -- for K3 (danbo_mlp16_pack, once per weight update: the 74 forward chunks) and for the fused trunk of
// the TRAINING step (danbo_trunk_pack, once per optimizer step: forward + backward chunks):
//   k_trunk_prep  W_fv = W_v[:, :256] W_f (fp64 dot products), b_eff = b_v + W_v[:, :256] b_f, max |w| of the nine matrices
//   k_trunk_pack  74 forward chunks + 76 chunks of the input-gradient chain (k_mlp16_bwd.hip: the same matrices
//                 transposed), every matrix times the power of two that puts its largest entry into [2^13, 2^14): unscaled, the
//                 lo half of every weight below 2^-3 is an fp16 SUBNORMAL (absolute error 2^-25 whatever the weight -- the largest
//                 single contribution to the split products' distance from fp32, tools/diag/f16split_emulation.py); the consumer's
//                 epilogue multiplies by the exact inverse inside the fma that adds the bias
// Backward chunk order: W_fv^T (4) | W_7^T (8) | W_6^T (8) | W_5[:, :195]^T (8, PE-ordered output rows, 13 tiles) |
// W_5[:, 195:]^T (8) | W_4^T .. W_1^T (8 each) | W_0^T (8, PE-ordered output rows).
// PE-ordered rows: output tile T, row 4 q + i of the tile = d pe_j of lane group q's channel kk = q + 4 c, j = 4 T + i = 13 c + t
// -- the slot order in which the forward's lanes hold the encoding (pe_kstep), so the adjoint of the encoding is lane-local.
// ---------------------------------------------------------------------------------------------
constexpr int NCH_BWD = 4 + 2 * NCH_ACT + 2 * NCH_ACT + 4 * NCH_ACT + NCH_ACT;   // 76
static_assert(NCH_TOTAL == DANBO_TRUNK_FWD_CHUNKS && NCH_BWD == DANBO_TRUNK_BWD_CHUNKS, "header constants out of date");

struct TrunkPackArgs {
    const float* pts_w[8];
    const float *feature_w, *feature_b, *views_w, *views_b;
    int Cv;
    float *wfv, *b_eff, *wmax, *winv;
    _Float16* packed;
    int n_chunks;      // NCH_TOTAL (forward only: K3) or NCH_TOTAL + NCH_BWD
    int form;          // 16: fragments of v_mfma_f32_16x16x32_f16 (k_pe_mlp16, the training trunk); 32: of v_mfma_f32_32x32x16_f16 (k_mlp32.hip)
};

// grid: 8 x 32 workgroups for the max |w| of pts_linears.0..7, then 128 workgroups -- one per row n of W_fv, thread = column f
constexpr int PREP_MAX_WGS = 32;
__global__ __launch_bounds__(256) void k_trunk_prep(TrunkPackArgs a) {
    __shared__ float s_m[4];
    __shared__ float s_row[W_];
    float mx = 0.f;
    int m;
    if ((int)blockIdx.x < 8 * PREP_MAX_WGS) {
        m = blockIdx.x / PREP_MAX_WGS;
        const int part = blockIdx.x % PREP_MAX_WGS;
        const long n = (long)W_ * (m == 0 ? IN_CH : (m == 5 ? IN_CH + W_ : W_));
        for (long i = (long)part * blockDim.x + threadIdx.x; i < n; i += (long)PREP_MAX_WGS * blockDim.x) mx = fmaxf(mx, fabsf(a.pts_w[m][i]));
    } else {
        m = 8;
        const int n = blockIdx.x - 8 * PREP_MAX_WGS, f = threadIdx.x;
        const int ld = W_ + a.Cv;
        s_row[f] = a.views_w[(size_t)n * ld + f];
        __syncthreads();
        // W_fv[n, f] = sum_c W_v[n, c] W_f[c, f] in fp64: the merged matrix is as exact as its factors
        double acc = 0.0;
#pragma unroll 8
        for (int c = 0; c < W_; ++c) acc = fma((double)s_row[c], (double)a.feature_w[(size_t)c * W_ + f], acc);
        a.wfv[n * W_ + f] = (float)acc;
        mx = fabsf((float)acc);
        // b_eff[n] = b_v[n] + sum_c W_v[n, c] b_f[c]
        double pb = (double)s_row[f] * (double)a.feature_b[f];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) pb += __shfl_xor(pb, off, 64);
        __shared__ double s_pb[4];
        if ((threadIdx.x & 63) == 0) s_pb[threadIdx.x >> 6] = pb;
        __syncthreads();
        if (threadIdx.x == 0) a.b_eff[n] = (float)((double)a.views_b[n] + ((s_pb[0] + s_pb[1]) + (s_pb[2] + s_pb[3])));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        mx = fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3]));
        if (mx > 0.f) atomicMax(reinterpret_cast<unsigned*>(a.wmax + m), __builtin_bit_cast(unsigned, mx));
    }
}

// element [n][k] of matrix m (0..7: pts_linears.m.weight [256, K_m]; 8: W_fv [128, 256])
__device__ __forceinline__ float trunk_w(const TrunkPackArgs& a, int m, int n, int k) {
    if (m == 8) return a.wfv[n * W_ + k];
    const int K = m == 0 ? IN_CH : (m == 5 ? IN_CH + W_ : W_);
    return a.pts_w[m][(size_t)n * K + k];
}

__global__ __launch_bounds__(256) void k_trunk_pack(TrunkPackArgs a) {
    if (blockIdx.x == 0 && threadIdx.x < 9) {
        float s, inv;
        weight_pow2_scale(a.wmax[threadIdx.x], s, inv);
        a.winv[threadIdx.x] = inv;
    }
    const long total = (long)a.n_chunks * (CHUNK_BYTES / 2);
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        int chunk = (int)(idx / (CHUNK_BYTES / 2));
        const int within = (int)(idx % (CHUNK_BYTES / 2));
        const int piece = within >> 9, lane = (within >> 3) & 63, e = within & 7;
        const int q = lane >> 4, ma = lane & 15;
        int mat;
        float w = 0.f;
        if (chunk < NCH_TOTAL) {            // ---------------- forward (K3's order)
            int layer, cl, kind;            // kind 0: x0 part, 1: act part, 2: view layer
            if (chunk < 7) { layer = 0; cl = chunk; kind = 0; }
            else if (chunk < 39) { layer = 1 + (chunk - 7) / 8; cl = (chunk - 7) % 8; kind = 1; }
            else if (chunk < 46) { layer = 5; cl = chunk - 39; kind = 0; }
            else if (chunk < 54) { layer = 5; cl = chunk - 46; kind = 1; }
            else if (chunk < 70) { layer = 6 + (chunk - 54) / 8; cl = (chunk - 54) % 8; kind = 1; }
            else { layer = 8; cl = chunk - 70; kind = 2; }
            mat = layer;
            if (a.form == 32) {
                // k_mlp32.hip: a chunk = two k-substeps of 16 (view layer: four) x 8 (4) output tiles of 32 x (hi, lo); lane (m, g = lane / 32)
                // holds row 32 T + m, k-slots 8 g .. 8 g + 7 of k-substep U.  Slot -> input feature: the order in which the lanes of
                // a 32 x 32 result tile hold the previous layer's outputs (feature 16 U + 8 (e / 4) + 4 g + e % 4), resp. the
                // encoding's slot j = 8 U + e = 13 c + t of channel 8 g + c
                const int g = lane >> 5, m32 = lane & 31;
                int U, T;
                if (kind == 2) { U = 4 * cl + (piece >> 3); T = (piece >> 1) & 3; }
                else { U = 2 * cl + (piece >> 4); T = (piece >> 1) & 7; }
                const int n = 32 * T + m32;
                if (kind == 0) {
                    const int j = 8 * U + e, c = j / 13, t = j % 13, kk = 8 * g + c;
                    if (j < 104 && kk < FEAT) w = trunk_w(a, layer, n, FEAT * t + kk);
                } else {
                    const int f = 16 * U + 8 * (e >> 2) + 4 * g + (e & 3);
                    w = trunk_w(a, layer, n, layer == 5 ? IN_CH + f : f);
                }
            } else {
            int s, T;
            if (kind == 2) { s = 2 * cl + (piece >> 4); T = (piece >> 1) & 7; }
            else { s = cl; T = piece >> 1; }
            const int n = 16 * T + ma;
            if (kind == 0) {
                const int j = 8 * s + e;
                const int c = j / 13, t = j % 13, kk = q + 4 * c;
                if (j < 52 && kk < FEAT) w = trunk_w(a, layer, n, FEAT * t + kk);
            } else {
                const int f = 16 * (2 * s + (e >> 2)) + 4 * q + (e & 3);
                w = trunk_w(a, layer, n, layer == 5 ? IN_CH + f : f);
            }
            }
        } else {                            // ---------------- backward: transposed
            chunk -= NCH_TOTAL;
            int s, kind;                    // kind 0: plain output rows, 1: PE-ordered output rows
            int koff = 0;                   // first input column of the forward matrix this GEMM's output rows index
            if (chunk < 4) { mat = 8; s = chunk; kind = 0; }
            else if (chunk < 20) { mat = 7 - (chunk - 4) / 8; s = (chunk - 4) % 8; kind = 0; }
            else if (chunk < 28) { mat = 5; s = chunk - 20; kind = 1; }
            else if (chunk < 36) { mat = 5; s = chunk - 28; kind = 0; koff = IN_CH; }
            else if (chunk < 68) { mat = 4 - (chunk - 36) / 8; s = (chunk - 36) % 8; kind = 0; }
            else { mat = 0; s = chunk - 68; kind = 1; }
            const int T = piece >> 1;
            const int f = 16 * (2 * s + (e >> 2)) + 4 * q + (e & 3);      // the GEMM's input feature = the forward layer's output
            if (kind == 0) w = trunk_w(a, mat, f, koff + 16 * T + ma);
            else if (T < 13) {
                const int j = 4 * T + (ma & 3), c = j / 13, t = j % 13, kk = (ma >> 2) + 4 * c;
                if (kk < FEAT) w = trunk_w(a, mat, f, FEAT * t + kk);
            }
        }
        float sc, inv;
        weight_pow2_scale(a.wmax[mat], sc, inv);
        w *= sc;
        const _Float16 hi = (_Float16)w;
        a.packed[idx] = (piece & 1) ? (_Float16)(w - (float)hi) : hi;
    }
}

// ---------------------------------------------------------------------------------------------
// the fused MLP
// ---------------------------------------------------------------------------------------------
struct Mlp16Args {
    const float* h;
    const int32_t* list;
    const int32_t* count;
    int n_cap;
    int S;
    const char* packed;
    const float* pts_b[8];
    const float* alpha_w;
    const float* alpha_b;
    const float* cview;
    const float* rgb_w;
    const float* rgb_b;
    float* raw_out;
    float* aux_out;
};

constexpr int M16_TABLE_FLOATS = 8 * W_ + W_ + 3 * VW_ + 4;
constexpr int M16_LDS_BYTES = RING_SLOTS * CHUNK_BYTES + M16_TABLE_FLOATS * 4 + 8 * (1024 + 256);  // + staging, see TileSrc

// one base address + instruction offsets: no per-load address registers.  Four column tiles per call.
template <int HALF>
__device__ __forceinline__ void prefetch_cv(const float* base, f32x4 (&cv)[8]) {
#define DANBO_CV_LOAD(T) asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(cv[T]) : "v"(base), "i"((T) * 64) : "memory")
    DANBO_CV_LOAD(4 * HALF); DANBO_CV_LOAD(4 * HALF + 1); DANBO_CV_LOAD(4 * HALF + 2); DANBO_CV_LOAD(4 * HALF + 3);
#undef DANBO_CV_LOAD
}

// ---------------------------------------------------------------------------------------------
// TRAIN instantiation (k_train_mlp_fwd): the same chain over the rows of one pass of a training step, with
//   * every matrix packed times a power of two (danbo_trunk_pack; the epilogue multiplies by the exact inverse `winv`),
//   * everything the backward pass and the weight-gradient kernel need written on the way, each value ONCE and in the order
//     the lanes hold it (fragment order: one contiguous KB per store instruction of a wavefront):
//       y_l   [rows, 256] l = 0..7  post-ReLU activations          (x operand of dW_{l+1}, of dW_fv / d alpha_w)
//       relu_l 64 bits per (row, lane group): [y_l > 0]            (mask of the input-gradient chain)
//       pe    [rows, 224] the 7 x 32 positional-encoding k-slots    (x operand of dW_0 and of the skip layer's dW)
//       hv    [rows, 128] post-ReLU view layer + its 32 sign bits   (x operand of d rgb_w; mask of d pre_v)
//       raw_rows [rows, 4] and the scatter into the dense raw of the pass / the ray's empty-space raw; row_ray [rows]
//   * rows = [0, R + n_c) in pass 0 and [tile boundary at or below R + n_c, R + n_c + n_f) in pass 1 (the rows of the first
//     tile that belong to pass 0 are recomputed, bit for bit: the fragment-order buffers are written in whole row groups).
// ---------------------------------------------------------------------------------------------
struct TrainFwd {
    int32_t* cnt;               // device counters of the step (k_train_rows.hip): read [0], [1]; block 0 derives the others
    const int32_t* row_sample;  // [rows]: sample index of row i >= R inside its pass
    int pass, R, S_c, S_f;
    const float* winv;          // [9]: exact inverses of the pack scales of pts_linears.0..7 and W_fv
    float* y; long y_stride;    // y_l = y + l * y_stride (floats)
    float* pe;
    unsigned long long* relu; long relu_stride;
    float* hv;
    unsigned* hv_bits;
    float *raw_rows, *raw_c, *raw_f, *raw_empty;
    int32_t* row_ray;
};

// B fragments of k-step s of the next GEMM from the previous layer's accumulators: tiles 2s and 2s+1,
// bias + ReLU, hi/lo split.  ALPHA: also accumulate this lane's part of the density logit.
// acc * winv + bias (one fma).  TRAIN: the eight activations are stored (ybase: wave-uniform address of this k-step's 2 KB)
// and their signs as byte S of the lane's 8 bytes at rbase (bit e of byte S <-> bit 8 S + e of the 64-bit word the
// input-gradient chain loads: column 16 (2 S + e / 4) + 4 q + e % 4).
template <bool ALPHA, bool TRAIN = false, int S = 0>
__device__ __forceinline__ void act_fragment(const f32x4& a0, const f32x4& a1, const float* bias /* + 4qq + 32s */,
                                             const float* aw, float& alpha_part, half8& bh, half8& bl, float winv = 1.f,
                                             const float* ybase = nullptr, const void* rbase = nullptr) {
    float v[8];
    // the tables are read with untracked ds_read (lds_table_read2): a read the compiler tracks makes it wait for every LDS-DMA
    // load of the weight ring in flight first (it cannot tell the tables from the ring), i.e. drain the ring once per k-step
    float4 b0, b1;
    lds_table_read2(bias, b0, b1);
    // acc * winv (the exact inverse of the matrix' pack scale, a power of two) + bias in ONE fma: the scale costs no instruction
    v[0] = fmaxf(fmaf(a0[0], winv, b0.x), 0.f); v[1] = fmaxf(fmaf(a0[1], winv, b0.y), 0.f);
    v[2] = fmaxf(fmaf(a0[2], winv, b0.z), 0.f); v[3] = fmaxf(fmaf(a0[3], winv, b0.w), 0.f);
    v[4] = fmaxf(fmaf(a1[0], winv, b1.x), 0.f); v[5] = fmaxf(fmaf(a1[1], winv, b1.y), 0.f);
    v[6] = fmaxf(fmaf(a1[2], winv, b1.z), 0.f); v[7] = fmaxf(fmaf(a1[3], winv, b1.w), 0.f);
    if (ALPHA) {
        float4 w0, w1;
        lds_table_read2(aw, w0, w1);
        alpha_part = fmaf(v[0], w0.x, alpha_part); alpha_part = fmaf(v[1], w0.y, alpha_part);
        alpha_part = fmaf(v[2], w0.z, alpha_part); alpha_part = fmaf(v[3], w0.w, alpha_part);
        alpha_part = fmaf(v[4], w1.x, alpha_part); alpha_part = fmaf(v[5], w1.y, alpha_part);
        alpha_part = fmaf(v[6], w1.z, alpha_part); alpha_part = fmaf(v[7], w1.w, alpha_part);
    }
    if (TRAIN) {
        const unsigned l16 = lane_off16();
        store16_s<0>(ybase, l16, f32x4{v[0], v[1], v[2], v[3]});
        store16_s<1024>(ybase, l16, f32x4{v[4], v[5], v[6], v[7]});
        // v >= 0: its bit pattern is non-zero exactly where the activation is positive
        unsigned w = 0u;
#pragma unroll
        for (int e = 0; e < 8; ++e) w |= min(__builtin_bit_cast(unsigned, v[e]), 1u) << e;
        asm volatile("global_store_byte %0, %1, %2 offset:%3" ::"v"(l16 >> 1), "v"(w), "s"(rbase), "i"(S) : "memory");
    }
    split8(v, bh, bl);
}

// the 8 positional-encoding values of k-step KS held by this lane: value j = 8 KS + e = 13 c + t of channel c
// (hv[c]): t = 0: x_c, t = 1 + 2l: sin(2^l x_c), t = 2 + 2l: cos(2^l x_c); j >= 52 is padding.  cs_keep carries the
// cosine of a sine / cosine pair across a k-step boundary.  (Deriving odd levels from the level below by the
// double-angle identities is 40 % cheaper but its 4e-7 input error is amplified past the 1e-4 bound by this MLP.)
template <int KS>
__device__ __forceinline__ void pe_kstep(const float (&hv)[4], float& cs_keep, float (&v8)[8]) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int j = 8 * KS + e, c = j / 13, t = j % 13;
        float val = 0.f;
        if (j < 52) {
            if (t == 0) val = hv[c];
            else if (t & 1) {
                float sn;
                pe_sincos(hv[c] * (float)(1 << ((t - 1) >> 1)), &sn, &cs_keep);
                val = sn;
            } else val = cs_keep;
        }
        v8[e] = val;
    }
}

template <bool TRAIN>
__device__ __forceinline__ void mlp16_body(const Mlp16Args& a, const TrainFwd& tr) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* s_bias = reinterpret_cast<float*>(smem + RING_SLOTS * CHUNK_BYTES);  // [8][256]
    float* s_aw = s_bias + 8 * W_;                                              // [256]
    float* s_rgbw = s_aw + W_;                                                  // [3][128]
    float* s_misc = s_rgbw + 3 * VW_;                                           // alpha_b, rgb_b[3]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 15, qq = lane >> 4;
    // exact inverses of the nine matrices' pack scales: the training step's own table, K3: the trailer of the packed buffer
    const float* winv_tab = TRAIN ? tr.winv : reinterpret_cast<const float*>(a.packed + (size_t)NCH_TOTAL * CHUNK_BYTES);
    for (int i = tid; i < 8 * W_; i += M16_THREADS) s_bias[i] = a.pts_b[i >> 8][i & 255];
    if (tid < W_) s_aw[tid] = a.alpha_w[tid];
    for (int i = tid; i < 3 * VW_; i += M16_THREADS) s_rgbw[i] = a.rgb_w[i];
    if (tid < 4) s_misc[tid] = tid == 0 ? a.alpha_b[0] : a.rgb_b[tid - 1];

    int n, tile0 = 0, first_f = 0;
    if (TRAIN) {
        // rows of this pass from the step's counters: cnt[0] = running number of in-volume samples (n_c after the coarse cull,
        // n_c + n_f after the second), cnt[1] = n_c (written by pass 0's block 0 below: pass 0 itself takes it from cnt[0])
        const int n_run = tr.cnt[0];
        const int n_c = tr.pass == 0 ? n_run : tr.cnt[1];
        first_f = tr.R + n_c;
        n = min(tr.R + n_run, a.n_cap);
        if (tr.pass != 0) tile0 = first_f / M16_BM;
        if (blockIdx.x == 0 && tid == 0) {
            if (tr.pass == 0) { tr.cnt[1] = n_run; tr.cnt[2] = tr.R + n_run; }
            else {
                tr.cnt[3] = n_run - n_c; tr.cnt[4] = tr.R + n_run; tr.cnt[5] = n_run;
                tr.cnt[6] = first_f & ~127; tr.cnt[7] = n_run - n_c + (first_f & 127);
            }
        }
    } else n = resolve_count(a.count, a.n_cap);
    // Rounds (round 5).  The rows are [16 * G0, n) in groups of 16 (one group = one wavefront's samples).  A FULL round gives every
    // workgroup 8 groups (a 128-row tile); what is left over -- `rem` groups, up to 8 * gridDim.x - 1 -- is spread over ALL
    // workgroups, `gpw` groups each, instead of filling a few workgroups' tiles: a wavefront's chain through the 74 chunks takes as
    // long whatever its tile holds, but the wavefronts of a sparse tile have their SIMD's matrix pipe to themselves (no partner:
    // ~half the time per chunk), and the wavefronts WITHOUT a group only keep the weight ring's hand-overs going (idle_tile) instead
    // of pushing zeros through the MFMAs.  The training step's 320 + 141 tiles were 2 + 1 rounds of 256 with the last one 25 % /
    // 55 % full; the frame's 5 229 tiles 20.4 rounds in 21.
    // (TRAIN: up to the 128-row boundary -- the backward and the weight-gradient kernel walk whole 128-row tiles of the fragment-order
    // buffers, and a padding row's activations must be this step's finite values, not what the memory held: 0 x NaN is NaN)
    const int G0 = tile0 * 8, G = TRAIN ? ((n + M16_BM - 1) / M16_BM) * 8 : (n + 15) >> 4;
    const int per_round = 8 * (int)gridDim.x;
    const int full = (G - G0) / per_round, rem = (G - G0) - full * per_round;
    const int gpw = (rem + (int)gridDim.x - 1) / (int)gridDim.x;                       // 0: no partial round
    const int my_rounds = full + ((int)blockIdx.x * gpw < rem ? 1 : 0);
    if (my_rounds == 0) return;
    // first group of this workgroup in round r, and how many of its wavefronts have one
    auto round_base = [&](int r) { return r < full ? G0 + (r * (int)gridDim.x + (int)blockIdx.x) * 8 : G0 + full * per_round + (int)blockIdx.x * gpw; };
    auto round_waves = [&](int r) { return r < full ? 8 : min(gpw, G - round_base(r)); };

    Pipe p;
    p.packed = a.packed; p.ring = smem; p.issue_chunk = 0; p.issue_slot = 0; p.cons_slot = 0; p.wave = wave; p.lane = lane;
    p.early = wave < 4;  // wavefronts w and w+4 of a workgroup share a SIMD
    pipe_issue<NCH_TOTAL>(p);
    pipe_issue<NCH_TOTAL>(p);
    pipe_issue<NCH_TOTAL>(p);
    // first tile of this workgroup: the same two staging loads (later tiles: issued during the previous view layer)
    TileSrc src;
    src.h = a.h; src.list = TRAIN ? tr.row_sample : a.list; src.dummy = a.packed; src.n = n;
    src.stage = smem + RING_SLOTS * CHUNK_BYTES + M16_TABLE_FLOATS * 4 + wave * STAGE_BYTES;
    src.next_row0 = (round_base(0) + wave) * 16;
    prefetch_rows(src, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // ring chunks 0-2 and the first rows (tables: the same barrier)
    __syncthreads();
    agroup_prefetch0(ring_lane_addr());               // group 0 of chunk 0 (slot 0): every later chunk is prefetched by its predecessor

    for (int rnd = 0; rnd < my_rounds; ++rnd) {
        const int grp_i = round_base(rnd) + wave;            // this wavefront's row group
        if (wave >= round_waves(rnd)) {                      // no group for this wavefront (partial round only: the last one)
            idle_tile<NCH_TOTAL>(p);
            continue;
        }
        // ------------------------------------------------------------------ inputs (staged during the previous tile)
        // lane constants of the tile's prologue, re-derived per tile (see the head below)
        int zero_t = 0;
        asm volatile("" : "+s"(zero_t));
        const int lane_t = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, (unsigned)zero_t));
        const int m = lane_t & 15, qq_t = lane_t >> 4;
        const int row0 = grp_i * 16 + m;
        const bool row_ok = row0 < n;
        const float* sh = reinterpret_cast<const float*>(src.stage) + m * DANBO_H_STRIDE + qq_t;
        const int staged_dst = reinterpret_cast<const int*>(src.stage + STAGE_H_BYTES)[m];
        int dst, ray;
        bool has_h = row_ok;
        if (TRAIN) {
            // rows [0, R): the ray's empty-space row (h = 0); [R, first_f): coarse samples; [first_f, n): importance samples
            const bool empty = row0 < tr.R;
            dst = row_ok ? (empty ? row0 : staged_dst) : -1;
            ray = 0;           // derived from dst behind the layer loop
            has_h = row_ok && !empty;
            asm volatile("" : "+v"(dst));
        } else {
            dst = row_ok ? (a.list ? staged_dst : row0) : -1;
            asm volatile("" : "+v"(dst));  // materialised now: the staging area is overwritten during this tile's view layer
            ray = 0;                       // = dst / S, formed in the head: ONE register (dst) crosses the layer loop, not the 64-bit
                                           // offset of the ray's view constants the compiler would make of it here (and spill)
        }
        // this lane's 4 channels: kk = qq + 4c  (kk = 15 is padding)
        float hv[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) hv[c] = has_h ? sh[4 * c] : 0.f;
        if (qq_t == 3) hv[3] = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) asm volatile("" : "+v"(hv[c]));
        float alpha_part = 0.f;
        f32x4 prev[16];  // pre-bias outputs of the previous layer
        src.next_row0 = (round_base(rnd + 1) + wave) * 16;       // (clamped to the last row where there is no next round)
        // TRAIN: wave-uniform base addresses of this wavefront's row group (16 rows) in the fragment-order buffers
        const long grp = grp_i;
        // steps 0..7: the density trunk; step 8: the merged feature + view layer (128 outputs, tiles 0..7 of acc)
#pragma unroll 1
        for (int step = 0; step < 9; ++step) {
            f32x4 acc[16];
            if (step == 0 || step == 5) {  // input / skip connection: 7 k-steps of PE features
                // produced 8 values at a time right before the k-step that consumes them (and recomputed for the
                // skip connection rather than kept in 56 VGPRs across five layers).  Explicit k-steps: a single
                // `#pragma unroll` loop over the 56 values exceeds the unroll budget and turns into a runtime loop
                // with dynamically indexed registers.
                float cs_keep = 0.f;
                const float* peg = TRAIN ? tr.pe + grp * (16 * 224) : nullptr;
#define DANBO_X0_STEP(KS, FIRST_)                                                  \
                {                                                                  \
                    float v8[8], hk[4] = {hv[0], hv[1], hv[2], hv[3]};             \
                    /* re-defined after the previous chunk: the sincos of later k-steps must not be hoisted (and spilled) */ \
                    asm volatile("" : "+v"(hk[0]), "+v"(hk[1]), "+v"(hk[2]), "+v"(hk[3]));                            \
                    pe_kstep<KS>(hk, cs_keep, v8);                                 \
                    if (TRAIN && step == 0) {                                      \
                        const unsigned l16 = lane_off16();                         \
                        store16_s<0>(peg + (KS) * 512, l16, f32x4{v8[0], v8[1], v8[2], v8[3]});      \
                        store16_s<1024>(peg + (KS) * 512, l16, f32x4{v8[4], v8[5], v8[6], v8[7]});   \
                    }                                                              \
                    half8 xh, xl;                                                  \
                    split8(v8, xh, xl);                                            \
                    /* TRAIN: layer 0 stores 2 per chunk (the wait may leave this chunk's and the previous one's in flight); the \
                       skip layer's pass over the same code stores nothing */     \
                    /* (KS == 6 drains the MFMAs also in the skip layer, whose act chunks follow: 21 idle cycles per tile) */ \
                    chunk_mfma2<NCH_TOTAL, 16, false, FIRST_, (KS) == 6, TRAIN ? ((KS) == 0 ? 6 : 8) : 4, NoExtra, 4>(         \
                        acc, p, ring_lane_addr(), xh, xl, xh, xl, NoExtra(), step != 0);                                  \
                }
                DANBO_X0_STEP(0, true)
                DANBO_X0_STEP(1, false)
                DANBO_X0_STEP(2, false)
                DANBO_X0_STEP(3, false)
                DANBO_X0_STEP(4, false)
                DANBO_X0_STEP(5, false)
                DANBO_X0_STEP(6, false)
#undef DANBO_X0_STEP
            }
            // lane constants are re-derived here instead of living in (or being spilled from) registers across
            // the layer loop: a scratch reload is a VMEM op and its wait would drain the weight ring
            int zero = 0;
            asm volatile("" : "+s"(zero));  // keeps the two mbcnt ops inside the loop (no hoist + spill)
            const int q4 = (int)((__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, (unsigned)zero)) >> 4) & 3) * 4;
            // prev holds layer step-1's accumulators: its pack scale; TRAIN: its activation buffer
            float winv = 1.f;
            const float* yg = nullptr;
            const unsigned long long* rg = nullptr;
            if (step != 0) winv = winv_tab[step - 1];
            if (TRAIN && step != 0) {
                yg = tr.y + (long)(step - 1) * tr.y_stride + grp * 4096;
                rg = tr.relu + (long)(step - 1) * tr.relu_stride + grp * 64;
            }
            if (step != 0 && step != 8) {
                const float* bias = s_bias + (step - 1) * W_ + q4;
                {
                    half8 bh, bl;
                    act_fragment<false, TRAIN, 0>(prev[0], prev[1], bias, nullptr, alpha_part, bh, bl, winv, yg, rg);
                    // TRAIN: 3 stores here (two halves of a k-step + its sign byte) and >= 2 in the previous chunk -- except behind the
                    // skip layer's PE chunks, which store nothing
                    if (step == 5) chunk_mfma2<NCH_TOTAL, 16, false, false, false, TRAIN ? 7 : 4>(acc, p, ring_lane_addr(), bh, bl, bh, bl);
                    else chunk_mfma2<NCH_TOTAL, 16, false, true, false, TRAIN ? 9 : 4>(acc, p, ring_lane_addr(), bh, bl, bh, bl);
                }
#define DANBO_ACT_STEP(s)                                                                                               \
                {                                                                                                       \
                    half8 bh, bl;                                                                                       \
                    int off_s = 32 * (s);                                                                               \
                    asm volatile("" : "+v"(off_s));   /* no hoisting of all eight bias loads (the offset, not the pointer: */ \
                    const float* bias_s = bias + off_s;  /* the pointer must stay an LDS pointer) */                     \
                    act_fragment<false, TRAIN, (s)>(prev[2 * (s)], prev[2 * (s) + 1], bias_s, nullptr, alpha_part, bh, bl, winv,       \
                                                        yg + (s) * 512, rg);                                            \
                    chunk_mfma2<NCH_TOTAL, 16, false, false, (s) == 7, TRAIN ? 10 : 4>(acc, p, ring_lane_addr(), bh, bl, bh, bl); \
                }
                DANBO_ACT_STEP(1) DANBO_ACT_STEP(2) DANBO_ACT_STEP(3) DANBO_ACT_STEP(4) DANBO_ACT_STEP(5) DANBO_ACT_STEP(6) DANBO_ACT_STEP(7)
#undef DANBO_ACT_STEP
            }
            if (step == 8) {
                // view layer; while it runs: stage the next tile's rows (chunk 0)
                const float* bias = s_bias + 7 * W_ + q4;
                const float* aw = s_aw + q4;
                f32x4 (&accv)[8] = *reinterpret_cast<f32x4 (*)[8]>(&acc[0]);
#define DANBO_VIEW_STEP(c)                                                                                              \
                {                                                                                                       \
                    half8 b0h, b0l, b1h, b1l;                                                                           \
                    int off_c = 64 * (c);                                                                               \
                    asm volatile("" : "+v"(off_c));                                                                     \
                    const float* bias_c = bias + off_c;                                                                 \
                    const float* aw_c = aw + off_c;                                                                     \
                    act_fragment<true, TRAIN, 2 * (c)>(prev[4 * (c)], prev[4 * (c) + 1], bias_c, aw_c, alpha_part, b0h, b0l, winv,     \
                                                        yg + (2 * (c)) * 512, rg);                                      \
                    act_fragment<true, TRAIN, 2 * (c) + 1>(prev[4 * (c) + 2], prev[4 * (c) + 3], bias_c + 32, aw_c + 32, alpha_part,   \
                                                            b1h, b1l, winv, yg + (2 * (c) + 1) * 512, rg);              \
                    /* TRAIN: 6 stores per chunk here; chunk 0 follows layer 7's last chunk (3 stores) */               \
                    if ((c) == 0) chunk_mfma2<NCH_TOTAL, 8, true, true, false, TRAIN ? 13 : 4, StageRows>(accv, p, ring_lane_addr(), b0h, b0l, b1h, b1l, StageRows{src, lane});  \
                    else if ((c) == 1) chunk_mfma2<NCH_TOTAL, 8, true, false, false, TRAIN ? 18 : 6>(accv, p, ring_lane_addr(), b0h, b0l, b1h, b1l);   \
                    else chunk_mfma2<NCH_TOTAL, 8, true, false, (c) == 3, TRAIN ? 16 : 4>(accv, p, ring_lane_addr(), b0h, b0l, b1h, b1l); \
                }
                DANBO_VIEW_STEP(0) DANBO_VIEW_STEP(1) DANBO_VIEW_STEP(2) DANBO_VIEW_STEP(3)
#undef DANBO_VIEW_STEP
#pragma unroll
                for (int T = 8; T < 16; ++T) acc[T] = acc[T - 8];  // defined values for the copy below (never read)
            }
#pragma unroll
            for (int T = 0; T < 16; ++T) prev[T] = acc[T];
        }
        f32x4 (&accv)[8] = *reinterpret_cast<f32x4 (*)[8]>(&prev[0]);
        // the head's lane constants are re-derived HERE (two mbcnt operations): computed once at the top of the kernel they live --
        // i.e. are spilled and reloaded, each reload a VMEM operation whose wait drains the weight ring -- across the layer loop
        int zero_h = 0;
        asm volatile("" : "+s"(zero_h));
        const int lane_h = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, (unsigned)zero_h));
        const int qq = lane_h >> 4, lane = lane_h;
        const int row = grp_i * 16 + (lane_h & 15);               // as at the top of the tile
        asm volatile("" : "+v"(dst));
        if (TRAIN) ray = dst < 0 ? 0 : (row < tr.R ? dst : dst / (row < first_f ? tr.S_c : tr.S_f));
        else ray = dst >= 0 ? dst / a.S : 0;
        // this tile's per-ray view constants: eight untracked loads and ONE wait (a compiler-tracked load per column tile
        // would each wait with vmcnt(0)); the trunk's registers are free here
        f32x4 cvq[8];
        {
            const float* cvb = a.cview ? a.cview + (size_t)ray * VW_ + 4 * qq : reinterpret_cast<const float*>(a.packed);
            prefetch_cv<0>(cvb, cvq);
            prefetch_cv<1>(cvb, cvq);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int T = 0; T < 8; ++T) asm volatile("" : "+v"(cvq[T]));
        }
        // ------------------------------------------------------------------ colour head + output
        float pr = 0.f, pg = 0.f, pb = 0.f;
        float* aux = (a.aux_out && dst >= 0) ? a.aux_out + (size_t)row * (VW_ + 1) + 4 * qq : nullptr;
        const float winv_v = winv_tab[8];
        const float* hvg = TRAIN ? tr.hv + grp * 2048 : nullptr;
        unsigned hvb = 0u;
#pragma unroll
        for (int T = 0; T < 8; ++T) {
            const int nn = 16 * T;  // + 4*qq + i
            f32x4 c4 = cvq[T];
            if (!a.cview) c4 = f32x4{0.f, 0.f, 0.f, 0.f};
            // TRAIN keeps round 3's two roundings (pre = acc * winv, pre + c): its recorded fixtures; K3: one fma
            const float pre[4] = {accv[T][0] * winv_v, accv[T][1] * winv_v, accv[T][2] * winv_v, accv[T][3] * winv_v};
            if (aux) *reinterpret_cast<float4*>(aux + nn) = make_float4(pre[0], pre[1], pre[2], pre[3]);
            const float x[4] = {fmaxf(TRAIN ? pre[0] + c4[0] : fmaf(accv[T][0], winv_v, c4[0]), 0.f),
                                fmaxf(TRAIN ? pre[1] + c4[1] : fmaf(accv[T][1], winv_v, c4[1]), 0.f),
                                fmaxf(TRAIN ? pre[2] + c4[2] : fmaf(accv[T][2], winv_v, c4[2]), 0.f),
                                fmaxf(TRAIN ? pre[3] + c4[3] : fmaf(accv[T][3], winv_v, c4[3]), 0.f)};
            if (TRAIN) {
                if (T < 4) store16_s<0>(hvg + T * 256, (unsigned)lane * 16u, f32x4{x[0], x[1], x[2], x[3]});
                else store16_s<0>(hvg + 1024 + (T - 4) * 256, (unsigned)lane * 16u, f32x4{x[0], x[1], x[2], x[3]});
#pragma unroll
                for (int i = 0; i < 4; ++i) hvb |= min(__builtin_bit_cast(unsigned, x[i]), 1u) << (4 * T + i);
            }
            const float4 wr = *reinterpret_cast<const float4*>(s_rgbw + 0 * VW_ + nn + 4 * qq);
            const float4 wg = *reinterpret_cast<const float4*>(s_rgbw + 1 * VW_ + nn + 4 * qq);
            const float4 wb = *reinterpret_cast<const float4*>(s_rgbw + 2 * VW_ + nn + 4 * qq);
            pr = fmaf(x[0], wr.x, pr); pr = fmaf(x[1], wr.y, pr); pr = fmaf(x[2], wr.z, pr); pr = fmaf(x[3], wr.w, pr);
            pg = fmaf(x[0], wg.x, pg); pg = fmaf(x[1], wg.y, pg); pg = fmaf(x[2], wg.z, pg); pg = fmaf(x[3], wg.w, pg);
            pb = fmaf(x[0], wb.x, pb); pb = fmaf(x[1], wb.y, pb); pb = fmaf(x[2], wb.z, pb); pb = fmaf(x[3], wb.w, pb);
        }
        // combine the four lane groups (each holds a quarter of a sample's features)
        const float r_ = quad_sum(pr) + s_misc[1];
        const float g_ = quad_sum(pg) + s_misc[2];
        const float b_ = quad_sum(pb) + s_misc[3];
        const float al = quad_sum(alpha_part) + s_misc[0];
        if (TRAIN) {
            tr.hv_bits[grp * 64 + lane] = hvb;
            if (qq == 0 && dst >= 0) {
                const float4 r4 = make_float4(r_, g_, b_, al);
                reinterpret_cast<float4*>(tr.raw_rows)[row] = r4;
                float4* dense = row < tr.R ? reinterpret_cast<float4*>(tr.raw_empty) : (row < first_f ? reinterpret_cast<float4*>(tr.raw_c)
                                                                                                       : reinterpret_cast<float4*>(tr.raw_f));
                dense[dst] = r4;
                tr.row_ray[row] = ray;
            }
        } else if (qq == 0 && dst >= 0) {
            reinterpret_cast<float4*>(a.raw_out)[dst] = make_float4(r_, g_, b_, al);
            if (a.aux_out) a.aux_out[(size_t)row * (VW_ + 1) + VW_] = al;
        }
    }
    // every wavefront executed the same number of hand-overs; drain the ring (and the last chunk's fragment prefetch) before the
    // LDS is released
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}

// amdgpu_num_vgpr(224): v224..v255 belong to chunk_mfma2's fragment buffers (mlp16_core.hpp)
__global__ __launch_bounds__(M16_THREADS, 2) __attribute__((amdgpu_num_vgpr(224))) void k_pe_mlp16(Mlp16Args a) { mlp16_body<false>(a, TrainFwd{}); }
__global__ __launch_bounds__(M16_THREADS, 2) __attribute__((amdgpu_num_vgpr(224))) void k_train_mlp_fwd(Mlp16Args a, TrainFwd t) { mlp16_body<true>(a, t); }

}  // namespace danbo

using namespace danbo;

// trailer of the packed buffer behind the 74 chunks: winv [16] | wmax [16] | W_fv [128 x 256] (scratch of the pack kernels)
static int mlp_pack_form(int form, const float* const* pts_w, const float* feature_w, const float* feature_b,
                         const float* views_w, const float* views_b, int Cv, void* packed16,
                         float* views_b_eff, void* stream) {
    DANBO_CHECK_ARG(pts_w && feature_w && feature_b && views_w && views_b && packed16 && views_b_eff && Cv >= 0);
    TrunkPackArgs a;
    for (int i = 0; i < 8; ++i) { DANBO_CHECK_ARG(pts_w[i]); a.pts_w[i] = pts_w[i]; }
    a.feature_w = feature_w; a.feature_b = feature_b; a.views_w = views_w; a.views_b = views_b; a.Cv = Cv;
    float* trailer = reinterpret_cast<float*>(reinterpret_cast<char*>(packed16) + (size_t)NCH_TOTAL * CHUNK_BYTES);
    a.winv = trailer; a.wmax = trailer + 16; a.wfv = trailer + 32; a.b_eff = views_b_eff;
    a.packed = reinterpret_cast<_Float16*>(packed16);
    a.n_chunks = NCH_TOTAL;
    a.form = form;
    { const hipError_t e = hipMemsetAsync(a.wmax, 0, 16 * sizeof(float), (hipStream_t)stream); if (e != hipSuccess) return (int)e; }
    hipLaunchKernelGGL(k_trunk_prep, dim3(8 * PREP_MAX_WGS + VW_), dim3(256), 0, (hipStream_t)stream, a);
    // dev A/B (tools/micro_mlp16.py --noscale): maxima of 0 pack every matrix times 1 -- round 4's packing
    static const bool noscale = dev_env("DANBO_MLP16_NOSCALE", 0) != 0;
    if (noscale) { const hipError_t e = hipMemsetAsync(a.wmax, 0, 16 * sizeof(float), (hipStream_t)stream); if (e != hipSuccess) return (int)e; }
    hipLaunchKernelGGL(k_trunk_pack, dim3(2048), dim3(256), 0, (hipStream_t)stream, a);
    DANBO_LAUNCH_RET();
}

extern "C" int danbo_mlp16_pack(const float* const* pts_w, const float* feature_w, const float* feature_b,
                                 const float* views_w, const float* views_b, int Cv, void* packed16,
                                 float* views_b_eff, void* stream) {
    return mlp_pack_form(16, pts_w, feature_w, feature_b, views_w, views_b, Cv, packed16, views_b_eff, stream);
}

// the same 74 chunks in the fragment order of k_mlp32.hip
extern "C" int danbo_mlp32_pack(const float* const* pts_w, const float* feature_w, const float* feature_b,
                                 const float* views_w, const float* views_b, int Cv, void* packed32,
                                 float* views_b_eff, void* stream) {
    return mlp_pack_form(32, pts_w, feature_w, feature_b, views_w, views_b, Cv, packed32, views_b_eff, stream);
}

extern "C" int danbo_pe_mlp16_fwd(const float* h, const int32_t* list, const int32_t* count, int n, int S,
                                   const void* packed16, const float* const* pts_b, const float* alpha_w,
                                   const float* alpha_b, const float* cview, const float* rgb_w,
                                   const float* rgb_b, float* raw_out, float* aux_out, void* stream) {
    DANBO_CHECK_ARG(n >= 0 && S > 0 && h && packed16 && pts_b && raw_out);
    if (n == 0) return 0;
    Mlp16Args a;
    a.h = h; a.list = list; a.count = count; a.n_cap = n; a.S = S; a.packed = reinterpret_cast<const char*>(packed16);
    for (int i = 0; i < 8; ++i) a.pts_b[i] = pts_b[i];
    a.alpha_w = alpha_w; a.alpha_b = alpha_b; a.cview = cview;
    a.rgb_w = rgb_w; a.rgb_b = rgb_b; a.raw_out = raw_out; a.aux_out = aux_out;
    DANBO_ENSURE_LDS(k_pe_mlp16, M16_LDS_BYTES);
    // one workgroup per CU as soon as there is a row group (16 rows) for each: a partial round is spread over all of them
    const int groups = ceil_div(n, 16);
    const int grid = groups < num_cu() ? groups : num_cu();
    hipLaunchKernelGGL(k_pe_mlp16, dim3(grid), dim3(M16_THREADS), M16_LDS_BYTES, (hipStream_t)stream, a);
    DANBO_LAUNCH_RET();
}

// ---------------------------------------------------------------------------------------------
// fused trunk of the training step: C ABI (include/danbo_hip.h)
// ---------------------------------------------------------------------------------------------
extern "C" int danbo_trunk_pack(const DanboTrunkWeights* w, void* stream) {
    DANBO_CHECK_ARG(w && w->feature_w && w->feature_b && w->views_w && w->views_b && w->packed && w->wfv && w->b_eff && w->wmax && w->winv);
    DANBO_CHECK_ARG(w->view_ch >= 0);
    TrunkPackArgs a;
    for (int i = 0; i < 8; ++i) { DANBO_CHECK_ARG(w->pts_w[i]); a.pts_w[i] = w->pts_w[i]; }
    a.feature_w = w->feature_w; a.feature_b = w->feature_b; a.views_w = w->views_w; a.views_b = w->views_b; a.Cv = w->view_ch;
    a.wfv = w->wfv; a.b_eff = w->b_eff; a.wmax = w->wmax; a.winv = w->winv; a.packed = reinterpret_cast<_Float16*>(w->packed);
    a.n_chunks = NCH_TOTAL + NCH_BWD;
    a.form = 16;
    hipLaunchKernelGGL(k_trunk_prep, dim3(8 * PREP_MAX_WGS + VW_), dim3(256), 0, (hipStream_t)stream, a);
    hipLaunchKernelGGL(k_trunk_pack, dim3(2048), dim3(256), 0, (hipStream_t)stream, a);
    DANBO_LAUNCH_RET();
}

extern "C" int danbo_trunk_fwd(const DanboTrunkWeights* w, const DanboTrunkRows* r, int pass, void* stream) {
    DANBO_CHECK_ARG(w && r && (pass == 0 || pass == 1) && w->packed && w->winv && w->alpha_w && w->alpha_b && w->rgb_w && w->rgb_b);
    DANBO_CHECK_ARG(r->cnt && r->row_sample && r->h_rows && r->cview && r->R >= 0 && r->S > 0 && r->Sf > 0 && r->rows_cap >= r->R && r->rows_cap > 0);
    DANBO_CHECK_ARG(r->rows_pad >= (r->rows_cap + 127) / 128 * 128 && r->y && r->pe && r->relu && r->hv && r->hv_bits);
    DANBO_CHECK_ARG(r->raw_rows && r->raw_c && r->raw_f && r->raw_empty && r->row_ray);
    Mlp16Args a;
    a.h = r->h_rows; a.list = nullptr; a.count = nullptr; a.n_cap = r->rows_cap; a.S = r->S;
    a.packed = reinterpret_cast<const char*>(w->packed);
    for (int i = 0; i < 8; ++i) { DANBO_CHECK_ARG(w->pts_b[i]); a.pts_b[i] = w->pts_b[i]; }
    a.alpha_w = w->alpha_w; a.alpha_b = w->alpha_b; a.cview = r->cview; a.rgb_w = w->rgb_w; a.rgb_b = w->rgb_b;
    a.raw_out = nullptr; a.aux_out = nullptr;
    TrainFwd t;
    t.cnt = r->cnt; t.row_sample = r->row_sample; t.pass = pass; t.R = r->R; t.S_c = r->S; t.S_f = r->Sf; t.winv = w->winv;
    t.y = r->y; t.y_stride = r->rows_pad * 256; t.pe = r->pe;
    t.relu = reinterpret_cast<unsigned long long*>(r->relu); t.relu_stride = r->rows_pad * 4;
    t.hv = r->hv; t.hv_bits = r->hv_bits; t.raw_rows = r->raw_rows; t.raw_c = r->raw_c; t.raw_f = r->raw_f; t.raw_empty = r->raw_empty;
    t.row_ray = r->row_ray;
    DANBO_ENSURE_LDS(k_train_mlp_fwd, M16_LDS_BYTES);
    const int groups = ceil_div(r->rows_cap, 16);
    const int grid = groups < num_cu() ? groups : num_cu();
    hipLaunchKernelGGL(k_train_mlp_fwd, dim3(grid), dim3(M16_THREADS), M16_LDS_BYTES, (hipStream_t)stream, a, t);
    DANBO_LAUNCH_RET();
}

extern "C" int danbo_trunk_pe_column(int k) {
    if (k < 0 || k >= DANBO_TRUNK_PE_WIDTH) return -1;
    const int ks = k >> 5, e = 4 * ((k >> 4) & 1) + (k & 3), q = (k >> 2) & 3;
    const int j = 8 * ks + e, c = j / 13, t = j % 13, kk = q + 4 * c;
    return (j < 52 && kk < FEAT) ? FEAT * t + kk : -1;
}
