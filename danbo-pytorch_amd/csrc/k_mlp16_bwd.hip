// Input-gradient chain of the fused trunk of the training step: the mirror image of K3's register-resident forward
// (k_mlp16.hip).  Reference: what loss.backward() (core/trainer.py:563-576) propagates through NeRF.inference
// (core/networks/nerf.py:176-209) from d raw back to the blended feature h:
//     d pre_v = (d rgb W_rgb) * [hv > 0]                                  (VALU, 3 x 128 per row)
//     d y7    = d pre_v W_fv + d alpha w_alpha ;  dz7 = d y7 * [y7 > 0]    (W_fv = W_v[:, :256] W_f: feature + view merged)
//     d y_{l-1} = dz_l W_l ;  dz_{l-1} = d y_{l-1} * [y_{l-1} > 0]         l = 7 .. 1   (layer 5: [d pe | d y4] = dz5 W_5)
//     d pe    = dz0 W_0 + (skip part of layer 5) ;  d h = PE'(h)^T d pe
// One workgroup = 8 wavefronts x 16 rows; a wavefront's rows are the MFMA B operand and never leave registers: output tile T
// of a GEMM (lane (m, q): features 16 T + 4 q + i of row m) is the B fragment of the next GEMM's k-step T / 2.  The transposed
// weights (A operand, packed by danbo_trunk_pack) stream through the same 4-slot LDS ring as the forward's.
// Written on the way, once, in fragment order: dz_0 .. dz_7 and d pre_v (the gradient operands of the weight-gradient kernel
// k_dw16), d alpha, d raw of the in-volume rows, d h [rows, 16] for the K2 / K1b adjoint.
//
// Gradient range: gradients are ~1e-6 .. 1e-9, far below fp16.  Every wavefront pre-scales the B operand of each GEMM by
// the power of two that puts the largest |.| of ITS 16 rows into [8, 16) (a wave reduction over values it already holds) and
// multiplies the accumulators by the exact inverse -- per row group instead of per tensor as the layer-per-launch
// form did it, so no pass over the data and no dependence between workgroups.
#include "mlp16_core.hpp"

namespace danbo {

constexpr int B_NCH = DANBO_TRUNK_BWD_CHUNKS;      // 76
constexpr int BW_ = 256, BVW_ = 128;
constexpr int MB_TABLE_FLOATS = BW_ + 3 * BVW_ + 16;       // alpha_w, rgb_w, winv
// ring | tables | two staging areas per wavefront (the next tile's rows are fetched while this tile's are still needed: the
// encoding's adjoint re-reads h at the very end) | d h partials of the skip layer per wavefront | running maxima
constexpr int MB_STAGE_OFF = RING_SLOTS * CHUNK_BYTES + MB_TABLE_FLOATS * 4;
constexpr int MB_DH_OFF = MB_STAGE_OFF + 2 * 8 * STAGE_BYTES;
constexpr int MB_MAX_OFF = MB_DH_OFF + 8 * 4 * 64 * 4;
constexpr int MB_LDS_BYTES = MB_MAX_OFF + 8 * 16 * 4;
static_assert(MB_LDS_BYTES <= 160 * 1024, "LDS budget");

struct TrainBwd {
    const int32_t* cnt;
    const int32_t* row_sample;
    const float* h_rows;
    int R, n_cap;
    const char* packed;                       // the 76 backward chunks
    const float* winv;                        // [9]
    const float *alpha_w, *rgb_w;
    const unsigned long long* relu; long relu_stride;
    const unsigned* hv_bits;
    const f32x4 *d_raw_c, *d_raw_f;
    f32x4* d_raw_rows;
    float* dz; long dz_stride;
    float* dpre_v;
    f32x4* d_alpha4;
    float* d_h;
    float* maxabs;                            // [10]
};

// power of two s with max * s in [8, 16) (1 for max == 0 / non-finite) and its exact reciprocal
__device__ __forceinline__ void grad_pow2_scale(float maxabs, float& s, float& inv) {
    const unsigned E = (__builtin_bit_cast(unsigned, maxabs) >> 23) & 255u;
    unsigned se = (E == 0u || E == 255u) ? 127u : 257u - E;
    se = se < 1u ? 1u : (se > 253u ? 253u : se);
    s = __builtin_bit_cast(float, se << 23);
    inv = __builtin_bit_cast(float, (254u - se) << 23);
}

// product of two powers of two by exponent arithmetic (scalar ALU when both are wave-uniform)
__device__ __forceinline__ float pow2_mul(float a, float b) {
    return __builtin_bit_cast(float, __builtin_bit_cast(int, a) + __builtin_bit_cast(int, b) - 0x3f800000);
}

// max over the wavefront of a non-negative value, as a wave-uniform bit pattern (non-negative floats order like unsigned ints)
__device__ __forceinline__ unsigned wave_max_bits(float v) {
    float x = v;
#define DANBO_DPP_MAX(CTRL, ROWMASK)                                                                                         \
    x = fmaxf(x, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, ROWMASK, 0xf, false)));
    DANBO_DPP_MAX(0x111, 0xf) DANBO_DPP_MAX(0x112, 0xf) DANBO_DPP_MAX(0x114, 0xf) DANBO_DPP_MAX(0x118, 0xf)
    DANBO_DPP_MAX(0x142, 0xa) DANBO_DPP_MAX(0x143, 0xc)
#undef DANBO_DPP_MAX
    return (unsigned)__builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 63);
}

// k-step S of the next GEMM's B operand from the gradients in g0 / g1 (tiles 2 S, 2 S + 1, already in true units):
// ReLU adjoint by the recorded sign bits (bit 8 S + e of `bits`), optional store of the masked values (dz fragment), pre-scale, split
template <int S, bool STORE>
__device__ __forceinline__ void grad_fragment(const f32x4& g0, const f32x4& g1, const uint2& bits, float sc, const float* zbase, half8& bh,
                                              half8& bl) {
    float v[8];
    const unsigned word = S < 4 ? bits.x : bits.y;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int k = 8 * (S & 3) + e;
        const int keep = (int)(word << (31 - k)) >> 31;                     // all ones where the activation was positive
        v[e] = __builtin_bit_cast(float, __builtin_bit_cast(int, e < 4 ? g0[e] : g1[e - 4]) & keep);
    }
    if (STORE) {
        const unsigned l16 = lane_off16();
        store16_s<0>(zbase, l16, f32x4{v[0], v[1], v[2], v[3]});
        store16_s<1024>(zbase, l16, f32x4{v[4], v[5], v[6], v[7]});
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] *= sc;
    split8(v, bh, bl);
}

// adjoint of the positional encoding for this lane's four channels: acc tile T, register i holds d pe_j, j = 4 T + i = 13 c + t
// (t = 0: x; t = 1 + 2 l: sin(2^l x); t = 2 + 2 l: cos(2^l x)), in units of 1 / u.
__device__ __forceinline__ void pe_adjoint(const f32x4 (&acc)[16], float u, const float (&hv)[4], float (&dh)[4]) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        float g = acc[(13 * c) >> 2][(13 * c) & 3];
#pragma unroll
        for (int l = 0; l < 6; ++l) {
            float sn, cs;
            const float f = (float)(1 << l);
            pe_sincos(hv[c] * f, &sn, &cs);
            const int js = 13 * c + 1 + 2 * l, jc = js + 1;
            g = fmaf(f * cs, acc[js >> 2][js & 3], g);
            g = fmaf(-f * sn, acc[jc >> 2][jc & 3], g);
        }
        dh[c] = fmaf(g, u, dh[c]);
    }
}

struct StageRowsIf {      // pipe_handover's `extra`: the staging loads of the next tile, in the last GEMM only
    const TileSrc& t;
    bool on;
    __device__ __forceinline__ void operator()() const {
        if (on) prefetch_rows(t, (int)(lane_off16() >> 4));    // lane re-derived: no hoisted per-lane addresses to spill
    }
};

// amdgpu_num_vgpr(224): v224..v255 belong to chunk_mfma2's fragment buffers (mlp16_core.hpp)
__global__ __launch_bounds__(M16_THREADS, 2) __attribute__((amdgpu_num_vgpr(224))) void k_train_mlp_bwd(TrainBwd a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* s_aw = reinterpret_cast<float*>(smem + RING_SLOTS * CHUNK_BYTES);   // [256]
    float* s_rgbw = s_aw + BW_;                                                // [3][128]
    float* s_winv = s_rgbw + 3 * BVW_;                                         // [16]
    unsigned* s_max = reinterpret_cast<unsigned*>(smem + MB_MAX_OFF);   // [8][16]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 15, qq = lane >> 4;
    if (tid < BW_) s_aw[tid] = a.alpha_w[tid];
    for (int i = tid; i < 3 * BVW_; i += M16_THREADS) s_rgbw[i] = a.rgb_w[i];
    if (tid < 16) s_winv[tid] = tid < 9 ? a.winv[tid] : 1.f;

    const int n = min(a.cnt[4], a.n_cap);
    const int first_f = a.cnt[2];
    const int ntiles = (n + M16_BM - 1) / M16_BM;
    if ((int)blockIdx.x >= ntiles) return;

    Pipe p;
    p.packed = a.packed; p.ring = smem; p.issue_chunk = 0; p.issue_slot = 0; p.cons_slot = 0; p.wave = wave; p.lane = lane;
    p.early = wave < 4;
    pipe_issue<B_NCH>(p);
    pipe_issue<B_NCH>(p);
    pipe_issue<B_NCH>(p);
    TileSrc src;
    src.h = a.h_rows; src.list = a.row_sample; src.dummy = a.packed; src.n = n;
    char* const stage0 = smem + MB_STAGE_OFF + wave * 2 * STAGE_BYTES;     // this wavefront's two staging areas
    src.stage = stage0;
    float* const s_dh = reinterpret_cast<float*>(smem + MB_DH_OFF) + wave * 256;
    src.next_row0 = blockIdx.x * M16_BM + wave * 16;
    prefetch_rows(src, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    agroup_prefetch0(ring_lane_addr());          // group 0 of chunk 0; every later chunk is prefetched by its predecessor

    unsigned run_max[10];          // running max |.| of this wavefront per output tensor (bit patterns, SGPRs)
#pragma unroll
    for (int i = 0; i < 10; ++i) run_max[i] = 0u;

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        // ------------------------------------------------------------------ inputs
        const int row = tile * M16_BM + wave * 16 + m;
        const bool row_ok = row < n;
        const char* const cur_stage = src.stage;           // holds THIS tile's rows; the next tile's go to the other area
        src.stage = cur_stage == stage0 ? stage0 + STAGE_BYTES : stage0;
        const int staged = reinterpret_cast<const int*>(cur_stage + STAGE_H_BYTES)[m];
        const long grp = (long)tile * 8 + wave;
        // d raw of the row (rows < R: the ray's sum over its samples outside every volume), the view layer's sign bits, layer 7's
        f32x4 draw;
        unsigned hvb;
        uint2 bits;
        {
            // (d_raw_c == nullptr: the caller hands d raw per ROW -- torch.ops.danbo.pe_mlp)
            const f32x4* src_d = !row_ok ? a.d_raw_rows : ((row < a.R || a.d_raw_c == nullptr) ? a.d_raw_rows + row : (row < first_f ? a.d_raw_c + staged : a.d_raw_f + staged));
            const unsigned* src_hb = a.hv_bits + grp * 64;                               // wave-uniform bases + lane offsets
            const unsigned long long* src_b = a.relu + 7 * a.relu_stride + grp * 64;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(draw) : "v"(src_d) : "memory");
            asm volatile("global_load_dword %0, %1, %2" : "=v"(hvb) : "v"((unsigned)lane * 4u), "s"(src_hb) : "memory");
            asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(bits) : "v"((unsigned)lane * 8u), "s"(src_b) : "memory");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("" : "+v"(draw), "+v"(hvb), "+v"(bits));
        }
        if (!row_ok) draw = f32x4{0.f, 0.f, 0.f, 0.f};
        if (qq == 0 && row_ok) {
            if (row >= a.R && a.d_raw_c != nullptr) a.d_raw_rows[row] = draw;
            a.d_alpha4[row] = f32x4{draw[3], 0.f, 0.f, 0.f};
        }
        src.next_row0 = (tile + (int)gridDim.x) * M16_BM + wave * 16;

        f32x4 prev[16];
        float u = 1.f;                 // prev * u = the gradient in true units
        uint2 nbits = bits;
        // ================================================================== head (peeled: d raw and the view layer's bits die here)
        {
            // d pre_v = (d rgb W_rgb) * [hv > 0] in the fragment layout of the view layer's 8 output tiles
            f32x4 acc[16];
            int zero = 0;
            asm volatile("" : "+s"(zero));
            const int q4 = (int)((__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, (unsigned)zero)) >> 4) & 3) * 4;
            float mx = 0.f;
#pragma unroll
            for (int T = 0; T < 8; ++T) {
                const float4 wr = *reinterpret_cast<const float4*>(s_rgbw + 0 * BVW_ + 16 * T + q4);
                const float4 wg = *reinterpret_cast<const float4*>(s_rgbw + 1 * BVW_ + 16 * T + q4);
                const float4 wb = *reinterpret_cast<const float4*>(s_rgbw + 2 * BVW_ + 16 * T + q4);
                const float w3[4][3] = {{wr.x, wg.x, wb.x}, {wr.y, wg.y, wb.y}, {wr.z, wg.z, wb.z}, {wr.w, wg.w, wb.w}};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float g = fmaf(draw[2], w3[i][2], fmaf(draw[1], w3[i][1], draw[0] * w3[i][0]));
                    const int keep = (int)(hvb << (31 - (4 * T + i))) >> 31;
                    const float gm = __builtin_bit_cast(float, __builtin_bit_cast(int, g) & keep);
                    prev[T][i] = gm;
                    mx = fmaxf(mx, fabsf(gm));
                }
            }
            const float* pg = a.dpre_v + grp * 2048;
            const unsigned l16 = (unsigned)lane * 16u;
#pragma unroll
            for (int T = 0; T < 8; ++T) {
                if (T < 4) store16_s<0>(pg + T * 256, l16, prev[T]);
                else store16_s<0>(pg + 1024 + (T - 4) * 256, l16, prev[T]);
            }
            const unsigned mb = wave_max_bits(mx);
            run_max[8] = max(run_max[8], mb);
            run_max[9] = max(run_max[9], wave_max_bits(fabsf(draw[3])));
            float sc, inv;
            grad_pow2_scale(__builtin_bit_cast(float, mb), sc, inv);
            // W_fv^T: 4 k-steps over the 128 view features.  The eight d pre_v stores above are younger than the ring's loads:
            // the first two hand-overs may leave them in flight.
#define DANBO_B0_STEP(s, FIRST_, WAIT_)                                                                                 \
            {                                                                                                           \
                const float v[8] = {prev[2 * (s)][0] * sc, prev[2 * (s)][1] * sc, prev[2 * (s)][2] * sc, prev[2 * (s)][3] * sc,         \
                                    prev[2 * (s) + 1][0] * sc, prev[2 * (s) + 1][1] * sc, prev[2 * (s) + 1][2] * sc, prev[2 * (s) + 1][3] * sc};   \
                half8 bh, bl;                                                                                           \
                split8(v, bh, bl);                                                                                      \
                chunk_mfma2<B_NCH, 16, false, FIRST_, (s) == 3, WAIT_>(acc, p, ring_lane_addr(), bh, bl, bh, bl);       \
            }
            DANBO_B0_STEP(0, true, 12)
            DANBO_B0_STEP(1, false, 12)
            DANBO_B0_STEP(2, false, 4)
            DANBO_B0_STEP(3, false, 4)
#undef DANBO_B0_STEP
            // d y7 = d pre_v W_fv + d alpha w_alpha, in true units
            const float un = pow2_mul(a.winv[8], inv);
            const float da = draw[3];
#pragma unroll
            for (int T = 0; T < 16; ++T) {
                const float4 aw = *reinterpret_cast<const float4*>(s_aw + 16 * T + q4);
                prev[T][0] = fmaf(acc[T][0], un, aw.x * da); prev[T][1] = fmaf(acc[T][1], un, aw.y * da);
                prev[T][2] = fmaf(acc[T][2], un, aw.z * da); prev[T][3] = fmaf(acc[T][3], un, aw.w * da);
            }
        }
        // ================================================================== the chain
        // steps: 1: W_7^T | 2: W_6^T | 3: W_5[:, :195]^T (PE rows) | 4: W_5[:, 195:]^T | 5..8: W_4^T .. W_1^T | 9: W_0^T (PE rows)
#pragma unroll 1
        for (int step = 1; step < 10; ++step) {
            f32x4 acc[16];
            const bool pe_out = step == 3 || step == 9;
            // layer whose pre-activation gradient this step's B operand is: 7, 6, 5, 5, 4, 3, 2, 1, 0
            const int L = step <= 3 ? 8 - step : 9 - step;
            float sc = 1.f, inv = 1.f;
            // ---- B operand = dz_L: prev (d y_L in units of 1 / u) -> true units, this wavefront's pre-scale
            {
                if (step != 4) {       // step 4 re-uses step 3's values (both multiply dz_5)
#pragma unroll
                    for (int T = 0; T < 16; ++T) prev[T] *= u;
                    u = 1.f;
                    bits = nbits;
                    asm volatile("" : "+v"(bits));
                }
                float mx = 0.f;
#pragma unroll
                for (int T = 0; T < 16; ++T)
                    mx = fmaxf(fmaxf(mx, fmaxf(fabsf(prev[T][0]), fabsf(prev[T][1]))), fmaxf(fabsf(prev[T][2]), fabsf(prev[T][3])));
                const unsigned mb = wave_max_bits(mx);      // of the unmasked values: an upper bound of max |dz_L|
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    if (i == L) run_max[i] = max(run_max[i], mb);
                grad_pow2_scale(__builtin_bit_cast(float, mb), sc, inv);
            }
            // the sign bits the NEXT conversion needs (layer L - 1), one GEMM ahead: an untracked load, landed long before
            // it is used (every hand-over in between waits for younger loads)
            if (step != 3 && step != 9) {
                const unsigned long long* src_b = a.relu + (long)(L - 1) * a.relu_stride + grp * 64;     // wave-uniform: SGPR base
                asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(nbits) : "v"(lane_off16() >> 1), "s"(src_b) : "memory");
            }
            const float* zg = a.dz + (long)L * a.dz_stride + grp * 4096;
            if (pe_out) {
#define DANBO_PE_STEP(s, FIRST_)                                                                                        \
                {                                                                                                       \
                    half8 bh, bl;                                                                                       \
                    grad_fragment<(s), true>(prev[2 * (s)], prev[2 * (s) + 1], bits, sc, zg + (s) * 512, bh, bl);        \
                    /* 13 of the chunk's 16 tiles: 7 fragment groups, so the buffer parity alternates over these 8 chunks (mlp16_core.hpp) */ \
                    if ((s) == 0) chunk_mfma2<B_NCH, 16, false, FIRST_, false, 8, StageRowsIf, 8, 13, 0>(acc, p, ring_lane_addr(), bh, bl, bh, bl, StageRowsIf{src, step == 9}); \
                    else chunk_mfma2<B_NCH, 16, false, FIRST_, (s) == 7, 8, NoExtra, 8, 13, (s) & 1>(acc, p, ring_lane_addr(), bh, bl, bh, bl); \
                }
                DANBO_PE_STEP(0, true)
                DANBO_PE_STEP(1, false) DANBO_PE_STEP(2, false) DANBO_PE_STEP(3, false) DANBO_PE_STEP(4, false) DANBO_PE_STEP(5, false)
                DANBO_PE_STEP(6, false) DANBO_PE_STEP(7, false)
#undef DANBO_PE_STEP
            } else {
                const bool quiet = step == 4;      // no stores in this pass over dz_5
#define DANBO_DX_STEP(s, FIRST_, W0)                                                                                    \
                {                                                                                                       \
                    half8 bh, bl;                                                                                       \
                    if (quiet) grad_fragment<(s), false>(prev[2 * (s)], prev[2 * (s) + 1], bits, sc, nullptr, bh, bl);   \
                    else grad_fragment<(s), true>(prev[2 * (s)], prev[2 * (s) + 1], bits, sc, zg + (s) * 512, bh, bl);   \
                    chunk_mfma2<B_NCH, 16, false, FIRST_, (s) == 7, W0, NoExtra, 4>(acc, p, ring_lane_addr(), bh, bl, bh, bl, NoExtra(), quiet); \
                }
                DANBO_DX_STEP(0, true, 6)
                DANBO_DX_STEP(1, false, 8) DANBO_DX_STEP(2, false, 8) DANBO_DX_STEP(3, false, 8) DANBO_DX_STEP(4, false, 8)
                DANBO_DX_STEP(5, false, 8) DANBO_DX_STEP(6, false, 8) DANBO_DX_STEP(7, false, 8)
#undef DANBO_DX_STEP
            }
            // ---- this GEMM's accumulators are in units of (pack scale of its matrix) x (pre-scale of its B operand)
            const float un = pow2_mul(a.winv[L], inv);
            if (pe_out) {
                // this lane's four channels of h (kk = qq + 4 c; empty-space rows and the padding channel: 0), re-read from the
                // staging area; the skip layer's contribution waits in LDS for layer 0's
                const float* sh = reinterpret_cast<const float*>(cur_stage) + m * DANBO_H_STRIDE + qq;
                float hv[4], dh[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) hv[c] = (row_ok && row >= a.R) ? sh[4 * c] : 0.f;
                if (qq == 3) hv[3] = 0.f;
#pragma unroll
                for (int c = 0; c < 4; ++c) dh[c] = step == 3 ? 0.f : s_dh[64 * c + lane];
                pe_adjoint(acc, un, hv, dh);
                if (step == 3) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) s_dh[64 * c + lane] = dh[c];
                } else if (row_ok) {
                    float* out = a.d_h + (size_t)row * 16 + qq;
#pragma unroll
                    for (int c = 0; c < 4; ++c) out[4 * c] = (qq == 3 && c == 3) ? 0.f : dh[c];
                }
            } else {
#pragma unroll
                for (int T = 0; T < 16; ++T) prev[T] = acc[T];
                u = un;
            }
        }
    }
    // ------------------------------------------------------------------ running maxima: one atomic per workgroup and tensor
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (lane < 10) {
        unsigned v = 0u;
#pragma unroll
        for (int i = 0; i < 10; ++i)
            if (lane == i) v = run_max[i];
        s_max[wave * 16 + lane] = v;
    }
    __syncthreads();
    if (tid < 10) {
        unsigned v = 0u;
#pragma unroll
        for (int w = 0; w < 8; ++w) v = max(v, s_max[w * 16 + tid]);
        if (v != 0u) atomicMax(reinterpret_cast<unsigned*>(a.maxabs) + tid, v);
    }
}

}  // namespace danbo

using namespace danbo;

extern "C" int danbo_trunk_bwd(const DanboTrunkWeights* w, const DanboTrunkRows* r, void* stream) {
    DANBO_CHECK_ARG(w && r && w->packed && w->winv && w->alpha_w && w->rgb_w);
    DANBO_CHECK_ARG(r->cnt && r->row_sample && r->h_rows && r->R >= 0 && r->rows_cap >= r->R && r->rows_pad >= (r->rows_cap + 127) / 128 * 128);
    DANBO_CHECK_ARG(r->relu && r->hv_bits && ((r->d_raw_c && r->d_raw_f) || (!r->d_raw_c && !r->d_raw_f)) && r->d_raw_rows && r->dz && r->dpre_v);
    DANBO_CHECK_ARG(r->d_alpha4 && r->d_h && r->maxabs);
    TrainBwd a;
    a.cnt = r->cnt; a.row_sample = r->row_sample; a.h_rows = r->h_rows; a.R = r->R; a.n_cap = r->rows_cap;
    a.packed = reinterpret_cast<const char*>(w->packed) + (size_t)DANBO_TRUNK_FWD_CHUNKS * CHUNK_BYTES;
    a.winv = w->winv; a.alpha_w = w->alpha_w; a.rgb_w = w->rgb_w;
    a.relu = reinterpret_cast<const unsigned long long*>(r->relu); a.relu_stride = r->rows_pad * 4;
    a.hv_bits = r->hv_bits;
    a.d_raw_c = reinterpret_cast<const f32x4*>(r->d_raw_c); a.d_raw_f = reinterpret_cast<const f32x4*>(r->d_raw_f);
    a.d_raw_rows = reinterpret_cast<f32x4*>(r->d_raw_rows);
    a.dz = r->dz; a.dz_stride = r->rows_pad * 256; a.dpre_v = r->dpre_v; a.d_alpha4 = reinterpret_cast<f32x4*>(r->d_alpha4);
    a.d_h = r->d_h; a.maxabs = r->maxabs;
    DANBO_ENSURE_LDS(k_train_mlp_bwd, MB_LDS_BYTES);
    const int ntiles = ceil_div(r->rows_cap, M16_BM);
    const int grid = ntiles < num_cu() ? ntiles : num_cu();
    hipLaunchKernelGGL(k_train_mlp_bwd, dim3(grid), dim3(M16_THREADS), MB_LDS_BYTES, (hipStream_t)stream, a);
    DANBO_LAUNCH_RET();
}
