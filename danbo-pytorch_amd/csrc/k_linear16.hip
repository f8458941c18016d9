// One dense layer  Y = act([X1 | X2] W^T + b)  with fp32-accurate products on the half-precision matrix cores
// (the hi/lo split of k_mlp16.hip: x = fp16(x) + fp16(x - fp16(x)), product = hi*hi + hi*lo + lo*hi, fp32 accumulate).
// (Accuracy: 2^-22 relative per product, plus an absolute floor of 2^-25 per operand -- the lo part of |x| < 1/8 is an fp16
// subnormal.  Against float64, 200 random layer shapes: <= 3e-6 of sum |x||w| per output, 6e-6 for a K = 4 layer.)
// gfx950 only.  Used where activations have to exist in HBM anyway: the A-NeRF trunk (W = 448, inputs 432 / 880) and
// the GEMMs of the training step.
//
// Dataflow (the transposed formulation of k_mlp16.hip, one layer per launch):
//   Y^T [N x rows] = W [N x K] * X^T [K x rows]
//   A operand = weights: pre-packed in MFMA fragment order (danbo_linear16_pack), streamed once per 128-row tile through
//               a 4-slot x 32 KB LDS ring by global_load_lds; one chunk = one k-step of 32 inputs x 16 output tiles;
//               fragments are read from LDS one group (two tiles) ahead of the MFMAs that use them, across chunk boundaries too
//               (round 5: mlp16_core.hpp group_mfma -- pinned fragment registers v224 .. v255, in-place interleaved MFMAs;
//               448 -> 448 at 1 M rows 1.36 -> 1.33 ms, the A-NeRF frame 193.7 -> 189.2 ms on one box.  The half-chunk stagger of
//               the two wavefronts of a SIMD that K3 lives on was measured again on this form: 1.39 ms, it still does not pay here);
//   B operand = this wavefront's 16 rows: lane (n, q) loads the 8 inputs 32 s + 8 q .. + 7 of row n straight from HBM
//               (32 B per lane, 128 B contiguous per row and k-step), two k-steps ahead, and splits them in registers;
//   D         = up to 32 output tiles of 16 features x 16 rows in registers (128 VGPRs); lane (n, q) owns features
//               16 T + 4 q + i of row n, i.e. 16-byte pieces of the output row: bias + activation + one dwordx4 store.
// Everything that is not an MFMA -- ring refill, hi/lo split of the next k-step's rows, their next request -- is spread over
// the eight MFMA batches of a chunk, so that the matrix pipe only idles at the barrier of the ring hand-over.
// Algorithmic HBM bytes per row: 4 (K + N); flops 2 K N (x3 half-precision MFMA products).
// Measured (tools/micro_linear16.py, MI355X): 448 -> 448 on 262 144 rows 0.43 ms = 248 TFLOP/s fp32-equivalent (the fp32
// library GEMM + ReLU: 1.07 ms); s_memtime trace of a k-step (--trace): 2 x ~1 700 cycles of MFMA batches, 2 x ~500 at the
// hand-over barrier (the two wavefronts of a SIMD finish their halves of a chunk one after the other), ~450 loop / tile
// bookkeeping; the end of a row tile (drain + 28 stores per lane) costs ~15 000 cycles per 80 000.
#include <type_traits>
#include "common.hpp"
#include "mlp16_core.hpp"

namespace danbo {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int L16_CHUNK = 32768, L16_SLOTS = 4, L16_BM = 128, L16_THREADS = 512, L16_MAX_N = 512;
constexpr int L16_LDS_BYTES = L16_SLOTS * L16_CHUNK + L16_MAX_N * 4;
constexpr int L16_COLOR_LD = 256, L16_COLOR_LDS_BYTES = L16_LDS_BYTES + 3 * L16_COLOR_LD * 4;   // + rgb_linear.weight for the colour epilogue

// ---------------------------------------------------------------------------------------------------------------
// packing: chunk c = s * NH + hf holds k-step s of output tiles 16 hf .. 16 hf + 15:
//   [tile t][hi | lo][lane 64][8 halves],  lane (m, q) = W[16 T + m][col(s, 8 q + e)]
// k-steps 0 .. KS1-1 cover the K1 columns of X1, the rest the K2 columns of X2 (each part zero-padded to 32)
// ---------------------------------------------------------------------------------------------------------------
// frag (bit 0: the K1 part, bit 1: the K2 part): that input arrives in FRAGMENT ORDER (see Lin16Args::frag) -- k-slot 8 q + e of
// k-step s then carries input column 32 s + 16 (e / 4) + 4 q + e % 4 instead of 32 s + 8 q + e
// enc_L > 0: the K1 part is A-NeRF's 24 (1 + 2 L) + 72 density inputs, RECOMPUTED by the kernel from the encoder's table
// (danbo_linear16_fwd_enc): 14 k-steps -- as many as the 432 columns take when they are read -- whose slots carry the columns in
// the order lin_enc_column gives them
__host__ __device__ inline int lin_enc_column(int s, int kk, int L) {
    const int q = kk >> 3, e = kk & 7, dir0 = (1 + 2 * L) * 24;
    if (s < 12) {      // joints 2 s, 2 s + 1; 16 slots each: [raw, direction x, sin_0, cos_0, ..., sin_6, cos_6], 4 per lane quarter
        const int j = 2 * s + (e >> 2), blk = 4 * q + (e & 3);
        if (blk == 0) return j;
        if (blk == 1) return dir0 + 3 * j;
        const int l = (blk - 2) >> 1;
        if (l >= L) return -1;
        return ((blk & 1) ? 2 + 2 * l : 1 + 2 * l) * 24 + j;
    }
    const int j = 16 * (s - 12) + 4 * q + (e >> 1);                // k-steps 12, 13: direction (y, z) of 16 + 8 joints
    return j < 24 ? dir0 + 3 * j + 1 + (e & 1) : -1;
}
constexpr int L16_ENC_K = 448, L16_ENC_FLOATS = 144;

__global__ __launch_bounds__(256) void k_linear16_pack(const float* __restrict__ w, long sn, long sk, int N, int K1, int K2, int NH,
                                                       int frag, int enc_L, _Float16* __restrict__ packed) {
    const int KS1 = (K1 + 31) / 32, KS = KS1 + (K2 + 31) / 32;
    const long total = (long)KS * NH * (L16_CHUNK / 2);
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int chunk = (int)(idx / (L16_CHUNK / 2)), within = (int)(idx % (L16_CHUNK / 2));
        const int piece = within >> 9, lane = (within >> 3) & 63, e = within & 7;
        const int s = chunk / NH, hf = chunk % NH;
        const int n = 16 * (16 * hf + (piece >> 1)) + (lane & 15);
        const int kk_rows = 8 * (lane >> 4) + e, kk_frag = 16 * (e >> 2) + 4 * (lane >> 4) + (e & 3);
        int col = -1;
        const int k1_cols = enc_L > 0 ? 24 * (1 + 2 * enc_L) + 72 : K1;     // columns of w in front of the K2 part
        if (s < KS1) {
            const int kk = (frag & 1) ? kk_frag : kk_rows;
            if (enc_L > 0) col = lin_enc_column(s, kk_rows, enc_L);
            else if (32 * s + kk < K1) col = 32 * s + kk;
        } else {
            const int kk = (frag & 2) ? kk_frag : kk_rows;
            if (32 * (s - KS1) + kk < K2) col = k1_cols + 32 * (s - KS1) + kk;
        }
        const float v = (n < N && col >= 0) ? w[n * sn + col * sk] : 0.f;
        const _Float16 hi = (_Float16)v;
        packed[idx] = (piece & 1) ? (_Float16)(v - (float)hi) : hi;
    }
}

struct Lin16Args {
    const float* x1;
    const float* x2;
    int ld1, ld2, K1, K2;
    const char* packed;
    const float* bias;
    float* y;
    int ldy, N, act, M;
    const int32_t* count;
    // Activations in FRAGMENT ORDER (inference instantiations; bit 0: x1, bit 1: x2, bit 2: y).  A [rows, C] activation, C a
    // multiple of 32 and rows padded to the 128-row tile, is stored as [rows / 16][C / 32][2][64 lanes][4 floats]: lane (n, q)
    // of row group g finds the 8 inputs of k-step s in two contiguous 16-byte pieces, (g, s, 0, lane) and (g, s, 1, lane), which
    // are exactly what it holds of output tiles 2 s and 2 s + 1 of the layer that produced them (features 32 s + 16 h + 4 q + i).
    // Every load and store instruction of a wavefront then moves one contiguous KB (8 full cache lines) instead of sixteen
    // 64-byte pieces of sixteen rows -- the row-major epilogue drains at 16 B/clk per CU (s_memtime: 14 000 cycles per tile).
    int frag;
    int enc_L;         // FRAG & 8: x1 is the A-NeRF encoder table [rows, 144] (k_anerf.hip), K1 = 448 slots of recomputed inputs, L levels
    long long* trace;  // dev tool (tools/micro_linear16.py --trace): s_memtime stamps of one wavefront, or nullptr
    // FRAG & 16: A-NeRF's colour head as the EPILOGUE of its head layer (danbo_linear16_fwd_color): the layer's N = VW + 1 outputs
    // (VW view features, then the density logit) never reach memory -- see the epilogue below
    const float *c_w, *c_C, *c_table, *c_rgb_w, *c_rgb_b;
    const int64_t* c_cam;
    float* c_raw;
    int c_ncodes, c_Rtot, c_ray0, c_S, c_VW;
};

struct LinPipe {
    const char* src_lane;   // packed + wave * 4096 + lane * 16: this lane's 16 bytes of piece 4 * wave of chunk 0
    char* dst_wave;         // ring + wave * 4096
    int issue_chunk, issue_slot, cons_slot, nch, skip;
};

// every wavefront loads 4 of the 32 pieces of a chunk; the instruction offset advances the global and the LDS address alike
__device__ __forceinline__ void lin_issue(LinPipe& p) {
    const auto* src = (const __attribute__((address_space(1))) char*)(p.src_lane + (size_t)p.issue_chunk * L16_CHUNK);
    auto* dst = (__attribute__((address_space(3))) char*)(p.dst_wave + p.issue_slot * L16_CHUNK);
    __builtin_amdgcn_global_load_lds(src, dst, 16, 0, 0);
    __builtin_amdgcn_global_load_lds(src, dst, 16, 1024, 0);
    __builtin_amdgcn_global_load_lds(src, dst, 16, 2048, 0);
    __builtin_amdgcn_global_load_lds(src, dst, 16, 3072, 0);
    p.issue_chunk = p.issue_chunk + 1 == p.nch ? 0 : p.issue_chunk + 1;
    p.issue_slot = p.issue_slot + 1 == L16_SLOTS ? 0 : p.issue_slot + 1;
}

// Ring hand-over before chunk c: my share of chunk c+1 has landed, then (barrier) everybody's has and everybody is done
// with chunk c-1, whose slot is refilled with chunk c+3 a little later, under chunk c's MFMAs (lin_issue).  vmcnt is an in-order counter for loads; the loads younger than
// those of chunk c+1 are the 4 of chunk c+2 plus the 2 input-row loads of one k-step, at every hand-over (the kernel
// issues the same loads in every k-step, past the end too), so "at most 6 outstanding" means chunk c+1 is in LDS -- and
// so are the input rows of the k-step that starts here, which were requested two k-steps ago (NH = 1: exactly the 6
// youngest loads are younger than them).  Stores in flight do not break this: the wait allows no more outstanding
// operations than there are younger LOADS.
template <class Stamp>
__device__ __forceinline__ void lin_handover(LinPipe& p, const Stamp& stamp) {
    if (p.skip > 0) --p.skip;                    // chunk c+1 is known to be there (prologue / end of a row tile)
    else __builtin_amdgcn_s_waitcnt(0xF76);      // vmcnt(6); expcnt, lgkmcnt: no wait
    stamp();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    stamp();
}

// (NOT common.hpp's three-instruction split8_mix: measured on this kernel it buys nothing -- 0.159 vs 0.160 ms at 256 -> 256)
__device__ __forceinline__ void lin_split8(const float (&v)[8], half8& hi, half8& lo) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const _Float16 hh = (_Float16)v[e];
        hi[e] = hh;
        lo[e] = (_Float16)(v[e] - (float)hh);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Input rows.  Lane (n, q) needs the 8 inputs 32 s + 8 q .. + 7 of its row in every k-step; they are requested two k-steps
// ahead.  A register that is the target of a load in flight must never be copied or spilled by the compiler, and across a
// loop back-edge it does exactly that with ordinary (even inline-asm "+v") values.  So the two row sets live in fixed
// physical registers v[208:215] / v[216:223] that only the inline asm below names: the kernel is compiled with
// amdgpu_num_vgpr(208), so the compiler never touches them (tests/test_host_logic.py checks the ISA), and
// it adds no waits of its own because it does not see the loads.  The ring hand-over's wait covers them (lin_handover).
// Columns past the end of a part are read from the clamped address, i.e. they repeat earlier columns of the same row:
// their packed weights are zero, so the (finite) values do not matter, nothing outside the row is touched, and every
// load instruction is issued for every lane in every k-step, which the vmcnt accounting relies on.
// ---------------------------------------------------------------------------------------------------------------
template <int SET>
__device__ __forceinline__ void lin_request_rows(const float* pa, const float* pb) {
    if (SET == 0) {
        asm volatile("global_load_dwordx4 v[208:211], %0, off\n\tglobal_load_dwordx4 v[212:215], %1, off" ::"v"(pa), "v"(pb)
                     : "memory", "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215");
    } else {
        asm volatile("global_load_dwordx4 v[216:219], %0, off\n\tglobal_load_dwordx4 v[220:223], %1, off" ::"v"(pa), "v"(pb)
                     : "memory", "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223");
    }
}
template <int SET>
__device__ __forceinline__ void lin_take_rows(float (&v)[8]) {
    if (SET == 0)
        asm volatile("v_mov_b32 %0, v208\n\tv_mov_b32 %1, v209\n\tv_mov_b32 %2, v210\n\tv_mov_b32 %3, v211\n\t"
                     "v_mov_b32 %4, v212\n\tv_mov_b32 %5, v213\n\tv_mov_b32 %6, v214\n\tv_mov_b32 %7, v215"
                     : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]), "=v"(v[4]), "=v"(v[5]), "=v"(v[6]), "=v"(v[7]));
    else
        asm volatile("v_mov_b32 %0, v216\n\tv_mov_b32 %1, v217\n\tv_mov_b32 %2, v218\n\tv_mov_b32 %3, v219\n\t"
                     "v_mov_b32 %4, v220\n\tv_mov_b32 %5, v221\n\tv_mov_b32 %6, v222\n\tv_mov_b32 %7, v223"
                     : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]), "=v"(v[4]), "=v"(v[5]), "=v"(v[6]), "=v"(v[7]));
}

// ---------------------------------------------------------------------------------------------------------------
// FRAG & 8: the inputs of the K1 part are not read, they are recomputed from the encoder's table (the 432-wide tensor the
// encoder would write is 1 728 B per row, written once and read by the first and the skip layer; the table is 576 B).  In a
// k-step s < 12 a lane's two 16-byte loads are the entries (inp, sh, w, dir x) of joints 2 s and 2 s + 1, and its eight inputs are
// slots 4 q .. 4 q + 3 of each joint: [inp w, dir x, sin(sh) w, cos(sh) w] for q = 0, else the sin / cos pairs of levels 2 q - 1 and
// 2 q times w (CutoffEmbedder._embed, cutoff_embedder.py:151-214; the arithmetic of k_anerf_encode with the sines from
// v_sin_f32 / v_cos_f32 after a two-term Cody-Waite reduction instead of sincosf: <= 2e-7 on the values).  k-steps 12, 13 carry
// the (y, z) components of the unit directions as stored.  14 k-steps: the MFMA work of the layer does not grow.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void lin_sincos(float a, float& sn, float& cs) {
    const float k = __builtin_rintf(a * 0.15915494309189535f);
    float r = __builtin_fmaf(k, -6.2831854820251465f, a);
    r = __builtin_fmaf(k, 1.7484555e-7f, r);                      // 2 pi = 6.2831854820251465 - 1.7484555e-7
    const float rev = r * 0.15915494309189535f;                   // |rev| <= 1/2
    sn = __builtin_amdgcn_sinf(rev);
    cs = __builtin_amdgcn_cosf(rev);
}
__device__ __forceinline__ void lin_enc_joint(float inp, float sh, float w, float dx, int q, int L, float* o) {
    const int l1 = q == 0 ? 0 : 2 * q - 1, l2 = 2 * q;
    float s1, c1, s2, c2;
    lin_sincos(sh * __builtin_bit_cast(float, (unsigned)(127 + l1) << 23), s1, c1);
    lin_sincos(sh * __builtin_bit_cast(float, (unsigned)(127 + l2) << 23), s2, c2);
    const bool on1 = l1 < L, on2 = l2 < L;
    o[0] = mul_rn(q == 0 ? inp : (on1 ? s1 : 0.f), w);
    o[1] = q == 0 ? dx : mul_rn(on1 ? c1 : 0.f, w);
    o[2] = mul_rn(on2 ? s2 : 0.f, w);
    o[3] = mul_rn(on2 ? c2 : 0.f, w);
}

// a group whose two tiles lie past N (known at run time only): no MFMAs, but the next group's fragments are still requested
template <int G>
__device__ __forceinline__ void lin_group_skip(unsigned cbase, unsigned nbase) {
    constexpr int NO = G == 7 ? 0 : (G + 1) * 4096;
    const unsigned nb = G == 7 ? nbase : cbase;
    if ((G & 1) == 0)
        asm volatile("s_waitcnt lgkmcnt(0)\n\tds_read_b128 v[240:243], %0 offset:%1\n\tds_read_b128 v[244:247], %0 offset:%2\n\t"
                     "ds_read_b128 v[248:251], %0 offset:%3\n\tds_read_b128 v[252:255], %0 offset:%4"
                     ::"v"(nb), "n"(NO), "n"(NO + 1024), "n"(NO + 2048), "n"(NO + 3072) : DANBO_A_CLOBBERS);
    else
        asm volatile("s_waitcnt lgkmcnt(0)\n\tds_read_b128 v[224:227], %0 offset:%1\n\tds_read_b128 v[228:231], %0 offset:%2\n\t"
                     "ds_read_b128 v[232:235], %0 offset:%3\n\tds_read_b128 v[236:239], %0 offset:%4"
                     ::"v"(nb), "n"(NO), "n"(NO + 1024), "n"(NO + 2048), "n"(NO + 3072) : DANBO_A_CLOBBERS);
}

// FRAG & 16 (the colour epilogue below): tiles T0 .. T0 + NT - 1 of acc += A_T . B with A_T = this lane's 8 gathered values of tile T
// (hi / lo split here), B = (wh, wl).  The gathers of the batch are requested together: one memory round trip per batch instead of
// one per tile (the compiler waits for ITS loads by count; nothing else is in flight at that point).  The MFMAs are inline asm like
// every MFMA of this kernel (a builtin makes the compiler budget AccVGPRs and leave the 208 registers it may use); `s_nop 1`: the
// operands come from VALU instructions the hazard recognizer cannot pair with an asm statement; the three products accumulate in
// place (back-to-back SrcC = vDst is interlocked by the hardware); MFMA_DRAIN: VALU instructions read the accumulators next.
template <int T0, int NT, int NACC>
__device__ __forceinline__ void lin_color_batch(f32x4 (&acc)[NACC], const float* const (&src)[8], int VW, int q, const half8& wh,
                                                const half8& wl) {
    // asm loads (immediate offsets on eight per-lane pointers) + ONE explicit wait: volatile asm statements keep their order, so at
    // most this batch's 8 NT values are in flight -- the compiler's own scheduler would hoist every gather of the epilogue to the top
    // (120 registers) and spill.  Tiles past VW (wave-uniform) are not loaded.
    float cv[NT][8];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if (16 * (T0 + t) < VW) {
#pragma unroll
            for (int e = 0; e < 8; ++e)
                asm volatile("global_load_dword %0, %1, off offset:%2" : "=v"(cv[t][e]) : "v"(src[e]), "n"(64 * (T0 + t)) : "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if (16 * (T0 + t) < VW) {
#pragma unroll
            for (int e = 0; e < 8; ++e) asm volatile("" : "+v"(cv[t][e]));      // defined by the loads above, not before
            // (lane quarter 3: k-slots 25 .. 31 carry the table value again -- finite -- against weights that are exactly 0)
            half8 ch, cl;
            split8_mix(cv[t], ch, cl);
            asm volatile("s_nop 1\n\t"
                         "v_mfma_f32_16x16x32_f16 %0, %1, %3, %0\n\t"
                         "v_mfma_f32_16x16x32_f16 %0, %1, %4, %0\n\t"
                         "v_mfma_f32_16x16x32_f16 %0, %2, %3, %0"
                         : "+v"(acc[T0 + t]) : "v"(ch), "v"(cl), "v"(wh), "v"(wl));
        }
    }
    asm volatile(DANBO_MFMA_DRAIN ::: "memory");
}

// NP: tile pairs in the last chunk of a k-step known at compile time (no branches in the batch loop), 0 = taken from N
// FRAG: Lin16Args::frag as a compile-time constant (a run-time choice costs the registers this kernel does not have)
// amdgpu_num_vgpr(208): v208 .. v223 belong to the row loads in flight, v224 .. v255 to group_mfma's fragment buffers
template <int NH, int NP, bool TRACE, int FRAG = 0>
__global__ __launch_bounds__(L16_THREADS, 1) __attribute__((amdgpu_num_vgpr(208))) void k_linear16(Lin16Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* s_bias = reinterpret_cast<float*>(smem + L16_SLOTS * L16_CHUNK);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, q = lane >> 4;
    const int M = resolve_count(a.count, a.M);
    const int n_tiles = (M + L16_BM - 1) / L16_BM;
    if ((int)blockIdx.x >= n_tiles) return;
    const int KS1 = (a.K1 + 31) / 32, KS = KS1 + (a.K2 + 31) / 32, nt = (a.N + 15) / 16;
    const int my_tiles = (n_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int G = my_tiles * KS;  // k-steps this workgroup runs

    for (int i = tid; i < L16_MAX_N; i += L16_THREADS) s_bias[i] = (a.bias != nullptr && i < a.N) ? a.bias[i] : 0.f;
    float* s_rgb = s_bias + L16_MAX_N;                      // FRAG & 16: rgb_linear.weight [3][256] (columns >= VW zero)
    if (FRAG & 16)
        for (int i = tid; i < 3 * L16_COLOR_LD; i += L16_THREADS) s_rgb[i] = (i % L16_COLOR_LD) < a.c_VW ? a.c_rgb_w[(i / L16_COLOR_LD) * a.c_VW + i % L16_COLOR_LD] : 0.f;

    LinPipe p{a.packed + wave * 4096 + lane * 16, smem + wave * 4096, 0, 0, 0, KS * NH, 2};
    lin_issue(p);
    lin_issue(p);
    lin_issue(p);

    // request the rows of the next k-step of this workgroup's stream (k-steps of its row tiles one after the other; past
    // the end the last k-step is requested again: the vmcnt pattern stays uniform)
    int rq_s = 0, rq_it = 0;
    auto request = [&](auto set) {
        long row = (long)(blockIdx.x + rq_it * gridDim.x) * L16_BM + wave * 16 + n;
        row = row < M ? row : M - 1;
        const bool second = rq_s >= KS1;
        if ((FRAG & 8) && !second) {
            // the encoder's table: entries of joints 2 s, 2 s + 1 (s < 12) / this lane quarter's four (y, z) pairs
            // (k-step 13 holds 8 joints: the upper two lane quarters re-read the lower ones' floats, their weights are zero)
            const int off = rq_s < 12 ? 8 * rq_s : 96 + 32 * (rq_s - 12) + 8 * ((rq_s == 13 && q >= 2) ? q - 2 : q);
            const float* tb = a.x1 + row * L16_ENC_FLOATS + off;
            lin_request_rows<decltype(set)::value>(tb, tb + 4);
        } else
        if (FRAG & (second ? 2 : 1)) {
            // fragment order: [row group][k-step][half][lane][4]
            const long g16 = (long)(blockIdx.x + rq_it * gridDim.x) * (L16_BM / 16) + wave;
            const int sp = second ? rq_s - KS1 : rq_s, ksp = (second ? a.K2 : a.K1) >> 5;
            const float* fb = (second ? a.x2 : a.x1) + ((g16 * ksp + sp) * 2) * 256 + lane * 4;
            lin_request_rows<decltype(set)::value>(fb, fb + 256);
        } else {
        const float* base = second ? a.x2 + row * a.ld2 : a.x1 + row * a.ld1;
        const int Kp = ((second ? a.K2 : a.K1) + 3) & ~3;   // rows are readable (and finite) up to a multiple of 4 columns
        const int col = 32 * (second ? rq_s - KS1 : rq_s) + 8 * q;
        lin_request_rows<decltype(set)::value>(base + min(col, Kp - 4), base + min(col + 4, Kp - 4));
        }
        if (rq_s + 1 < KS) ++rq_s;
        else if (rq_it + 1 < my_tiles) rq_s = 0, ++rq_it;
    };
    request(std::integral_constant<int, 0>{});
    request(std::integral_constant<int, 1>{});
    __builtin_amdgcn_s_waitcnt(0xF70);  // vmcnt(0): chunks 0-2 and the first rows are here (p.skip = 2 relies on it)
    __syncthreads();
    agroup_prefetch0((unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem + (unsigned)lane * 16u);   // group 0 of chunk 0
    half8 bh, bl;   // B fragments of the current k-step; those of the next one are prepared under this one's MFMAs
    {
        float x[8];
        lin_take_rows<0>(x);
        if (FRAG & 8) {                          // k-step 0 of the first row tile
            float o[8];
            lin_enc_joint(x[0], x[1], x[2], x[3], q, a.enc_L, o);
            lin_enc_joint(x[4], x[5], x[6], x[7], q, a.enc_L, o + 4);
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = o[e];
        }
        lin_split8(x, bh, bl);
        request(std::integral_constant<int, 0>{});
    }

    f32x4 acc[16 * NH];
    int s = 0, it = 0, tr = 0;
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem + (unsigned)lane * 16u;
    auto stamp = [&]() {   // compiled out unless the trace buffer is set (danbo_linear16_set_trace)
        if (TRACE && blockIdx.x == 0 && wave == 1 && it == 2 && lane == 0 && tr < 256) a.trace[tr++] = (long long)__builtin_amdgcn_s_memtime();
    };
    // one k-step (`set`: the register set holding the rows of k-step g + 1)
    auto kstep = [&](auto set, int g) {
        if (s == 0) {
#pragma unroll
            for (int T = 0; T < 16 * NH; ++T) acc[T] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        stamp();
        half8 nbh, nbl;
        float nx[8];
        // one chunk (k-step s, output tiles 16 hf .. 16 hf + 15) as eight hand-scheduled groups of two tiles (mlp16_core.hpp group_mfma:
        // round 5 of K3): A fragments in the pinned registers v224 .. v255, the reads of group b + 1 behind the first MFMA of group b,
        // the last group of a chunk reads group 0 of the NEXT chunk -- published by the hand-over in front of this one -- so a chunk's
        // MFMAs start right behind the barrier instead of an LDS round trip that all eight wavefronts make at the same moment.
        auto chunk = [&](auto hfc) {
            constexpr int hf = decltype(hfc)::value;
            // groups with MFMAs in this chunk when that is known at compile time (whole pairs of tiles past N are skipped)
            constexpr int NGC = (NP && hf == NH - 1) ? NP : 8;
            static_assert(NGC % 2 == 0, "an odd number of groups would leave the next chunk's fragments in the other buffer");
            stamp();
            lin_handover(p, stamp);
            stamp();
            const unsigned cbase = lds_base + (unsigned)(p.cons_slot * L16_CHUNK);
            p.cons_slot = p.cons_slot + 1 == L16_SLOTS ? 0 : p.cons_slot + 1;
            const unsigned nbase = lds_base + (unsigned)(p.cons_slot * L16_CHUNK);
            auto group = [&](auto bc) {
                constexpr int b = decltype(bc)::value;
                // Work that is not MFMA is spread over the groups, a few VALU / scalar instructions under each group's six
                // MFMAs: the ring refill (the slot of chunk c-1 is free once the barrier has been passed), and in the first
                // chunk of a k-step the next k-step's rows (requested two k-steps ago, arrived: see lin_handover) become B
                // fragments and their registers are re-used for the request after next.
                if (b == 1) lin_issue(p);
                if (hf == 0 && b == 2) lin_take_rows<decltype(set)::value>(nx);
                if ((FRAG & 8) && hf == 0 && (b == 3 || b == 4)) {
                    // rows of k-step s + 1 (0 behind the last one): the recomputed inputs, one joint under each of two groups
                    const int s_next = s + 1 < KS ? s + 1 : 0;
                    if (s_next < 12) {
                        float o[4];
                        constexpr int e0 = 4 * (b == 4);
                        lin_enc_joint(nx[e0], nx[e0 + 1], nx[e0 + 2], nx[e0 + 3], q, a.enc_L, o);
                        nx[e0] = o[0]; nx[e0 + 1] = o[1]; nx[e0 + 2] = o[2]; nx[e0 + 3] = o[3];
                    }
                }
                if ((FRAG & 8) ? (hf == 1 && b <= 3) : (hf == 0 && b >= 3 && b <= 6)) {
                    constexpr int b0 = (FRAG & 8) ? (b & 3) : ((b + 1) & 3);      // b resp. b - 3 where the condition holds
#pragma unroll
                    for (int e = 2 * b0; e < 2 * b0 + 2; ++e) {
                        const float xe = nx[e];
                        const _Float16 hh = (_Float16)xe;
                        nbh[e] = hh;
                        nbl[e] = (_Float16)(xe - (float)hh);
                    }
                }
                if (hf == 0 && b == 7) request(set);
                if (b < NGC) {
                    f32x4& c0 = acc[16 * hf + 2 * b];
                    f32x4& c1 = acc[16 * hf + 2 * b + 1];
                    if (NP || 16 * hf + 2 * b < nt) group_mfma<b, false, b == 0, false, NGC>(c0, c1, bh, bl, cbase, nbase);
                    else lin_group_skip<b>(cbase, nbase);     // N known at run time only: the reads keep the double buffer going
                }
            };
            group(std::integral_constant<int, 0>{}); group(std::integral_constant<int, 1>{});
            group(std::integral_constant<int, 2>{}); group(std::integral_constant<int, 3>{});
            group(std::integral_constant<int, 4>{}); group(std::integral_constant<int, 5>{});
            group(std::integral_constant<int, 6>{}); group(std::integral_constant<int, 7>{});
        };
        chunk(std::integral_constant<int, 0>{});
        if (NH == 2) chunk(std::integral_constant<int, NH - 1>{});
        bh = nbh;
        bl = nbl;
        stamp();
        if (++s < KS) return;
        // End of a row tile.  Everything requested so far -- chunks c+1, c+2 of the ring and the rows of the next two
        // k-steps, all at least a chunk old -- is waited for BEFORE the stores are issued, and the next two hand-overs then
        // skip their wait: a vmcnt wait right behind 28 stores per lane would stall every wavefront until the stores have
        // reached memory; this way they have two chunks of MFMA work to drain under.
        const long row = (long)(blockIdx.x + it * gridDim.x) * L16_BM + wave * 16 + n;
        __builtin_amdgcn_s_waitcnt(0xF70);
        p.skip = 2;
        // the MFMAs are inline asm: the wait states between the last of them and the VALU instructions that read the accumulators
        // are written out (the vmcnt wait above usually covers them, but does not have to), and every accumulator is re-defined
        // behind them so that no read is scheduled in front
        asm volatile(DANBO_MFMA_DRAIN ::: "memory");
#pragma unroll
        for (int T = 0; T < 16 * NH; ++T) asm volatile("" : "+v"(acc[T]));
        if (FRAG & 16) {
            // A-NeRF's colour head as the epilogue of its head layer (reference nerf.py:196-209; what k_anerf_color did on the
            // head's rows until round 5: 0.91 GB written and read back per 1 M-row chunk, 6 % of the frame).  The 16 rows of this
            // wavefront are 16 consecutive samples of ONE ray (S % 16 == 0, chunks of whole rays), so the view layer's
            // cutoff-weighted sum over the joints is one more k-step of the same GEMM:
            //     pre_v^T [VW x 16] += C_ray^T [VW x 24] . w^T [24 x 16]
            // A = the ray's joint vectors (lane (m, q): C[8 q + e][ray][16 T + m], gathered from [24][R][VW]: 64 contiguous bytes
            // per joint and tile), B = the rows' cutoff weights (lane (n, q): w[row n][8 q + e]), both split hi / lo like every
            // other operand; table[code(ray)] enters as k-slot 24 with weight 1.  Then ReLU, rgb_linear as 3 x 60 FMAs per lane and a sum over the four lane
            // quarters, and lane (n, 0) stores the sample's raw (rgb, density logit = feature VW of its own accumulators).
            const long row0g = (long)(blockIdx.x + it * gridDim.x) * L16_BM + wave * 16;       // wave-uniform
            if (row0g < M) {
                const int VW = a.c_VW, ray = a.c_ray0 + (int)(row0g / a.c_S);
                long code = a.c_ncodes;                                  // the mean code (Optcodes eval with idx < 0)
                if (a.c_cam) {
                    const long idx = a.c_cam[ray];
                    if (idx >= 0) code = idx < a.c_ncodes ? idx : a.c_ncodes - 1;
                }
                // ... and the table row rides along as a 25th "joint" with weight 1 (k-slot 24 = element 0 of lane quarter 3): its
                // loads are part of the same gathers instead of fifteen more round trips
                float wv[8];
                {
                    const float* wr = a.c_w + (row0g + n) * J + 8 * min(q, 2);
                    const f32x4 w0 = *reinterpret_cast<const f32x4*>(wr), w1 = *reinterpret_cast<const f32x4*>(wr + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { wv[e] = q < 3 ? w0[e] : 0.f; wv[4 + e] = q < 3 ? w1[e] : 0.f; }
                    if (q == 3) wv[0] = 1.0f;
                }
                half8 wh, wl;
                lin_split8(wv, wh, wl);
                const size_t jstride = q < 3 ? (size_t)a.c_Rtot * VW : 0;
                const float* Cb = q < 3 ? a.c_C + (size_t)ray * VW + n + (size_t)(8 * q) * jstride : a.c_table + (size_t)code * VW + n;
                // the gathers of eight (then seven) tiles are requested together: one memory round trip per batch instead of one per
                // tile (the compiler waits for ITS loads by count; nothing else is in flight here)
                const float* src[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) src[e] = Cb + (size_t)e * jstride;
                // (batches of five tiles: 40 gathers in flight.  v208 .. v255 hold the row loads and weight fragments of the NEXT k-step
                // across this epilogue and this toolchain does not enforce amdgpu_num_vgpr: the compiler has to stay below v208 by
                // itself -- 179 here, tests/test_host_logic.py checks the ISA of every build)
                lin_color_batch<0, 5>(acc, src, VW, q, wh, wl);
                lin_color_batch<5, 5>(acc, src, VW, q, wh, wl);
                lin_color_batch<10, 5>(acc, src, VW, q, wh, wl);
                float pr = 0.f, pg = 0.f, pb = 0.f;
                // ONE per-lane LDS address, compile-time offsets from it (left to itself the compiler keeps 60 hoisted addresses alive
                // across the k-step loop: the registers this epilogue needs)
                const float* lq = s_bias + 4 * q;
                asm volatile("" : "+v"(lq));
#pragma unroll
                for (int T = 0; T < 15; ++T) {
                    if (16 * T < VW) {
                        const f32x4 v = acc[T] + *reinterpret_cast<const f32x4*>(lq + 16 * T);
                        const f32x4 r0 = *reinterpret_cast<const f32x4*>(lq + L16_MAX_N + 16 * T),
                                    r1 = *reinterpret_cast<const f32x4*>(lq + L16_MAX_N + L16_COLOR_LD + 16 * T),
                                    r2 = *reinterpret_cast<const f32x4*>(lq + L16_MAX_N + 2 * L16_COLOR_LD + 16 * T);
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const float h = fmaxf(v[i], 0.f);
                            pr = fmaf(h, r0[i], pr);
                            pg = fmaf(h, r1[i], pg);
                            pb = fmaf(h, r2[i], pb);
                        }
                        if (T & 1) asm volatile("" ::: "memory");       // (two tiles' table reads in flight, not all fifteen: registers)
                    }
                }
                pr += lane_xor16(pr); pg += lane_xor16(pg); pb += lane_xor16(pb);
                pr += lane_xor32(pr); pg += lane_xor32(pg); pb += lane_xor32(pb);
                // the density logit: feature VW = element 0 of tile VW / 16 in the lanes of quarter 0
                float al = 0.f;
#pragma unroll
                for (int T = 0; T < 16; ++T)
                    if (16 * T == VW) al = acc[T][0] + s_bias[VW];
                if (q == 0)
                    reinterpret_cast<f32x4*>(a.c_raw)[(size_t)a.c_ray0 * a.c_S + row0g + n] =
                        f32x4{pr + a.c_rgb_b[0], pg + a.c_rgb_b[1], pb + a.c_rgb_b[2], al};
            }
        } else
        if (FRAG & 4) {
            // fragment-order output: tile T of this wavefront's 16 rows is one contiguous KB (rows past M: padding of the buffer)
            const long g16 = (long)(blockIdx.x + it * gridDim.x) * (L16_BM / 16) + wave;
            float* yf = a.y + g16 * (a.N >> 5) * 512 + lane * 4;
#pragma unroll
            for (int T = 0; T < 16 * NH; ++T) {
                if (16 * T < a.N) {
                    f32x4 v = acc[T] + *reinterpret_cast<const f32x4*>(s_bias + 16 * T + 4 * q);
                    if (a.act == 1) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.f);
                    }
                    asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(yf + 256 * T), "v"(v) : "memory");   // wait states: see mlp16_core.hpp store16_s
                }
            }
        } else
        if (row < M) {
            float* yrow = a.y + row * a.ldy;
#pragma unroll
            for (int T = 0; T < 16 * NH; ++T) {
                const int col = 16 * T + 4 * q;
                if (col < a.N) {
                    f32x4 v = acc[T] + *reinterpret_cast<const f32x4*>(s_bias + col);
                    if (a.act == 1) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.f);
                    }
                    // inline asm like the loads: no compiler-inserted waits.  The hand-over waits stay valid with stores in
                    // flight even if stores and loads complete out of order with respect to each other: they allow no more
                    // outstanding operations than there are younger LOADS, and loads return in order.
                    if (col + 4 <= a.N) asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(yrow + col), "v"(v) : "memory");
                    else {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            if (col + i < a.N) asm volatile("global_store_dword %0, %1, off" ::"v"(yrow + col + i), "v"(v[i]) : "memory");
                    }
                }
            }
        }
        stamp();
        s = 0;
        ++it;
    };
    // k-step g prepares the rows of k-step g + 1, which live in register set (g + 1) & 1
    for (int g = 0; g < G; g += 2) {
        kstep(std::integral_constant<int, 1>{}, g);
        if (g + 1 < G) kstep(std::integral_constant<int, 0>{}, g + 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // LDS-DMA still in flight must not outlive the workgroup's LDS
}

static inline int lin16_nh(int N) { return N <= 256 ? 1 : 2; }

}  // namespace danbo

using namespace danbo;

static long long* g_lin16_trace = nullptr;
/* dev tool: buffer of 256 int64 that receives s_memtime stamps of wavefront 1 of workgroup 0 during its third row tile */
extern "C" int danbo_linear16_set_trace(void* buf) {
    g_lin16_trace = (long long*)buf;
    return 0;
}

extern "C" int danbo_linear16_packed_bytes(int N, int K1, int K2) {
    if (N < 1 || N > L16_MAX_N || K1 < 1 || K2 < 0) return -1;
    return ((K1 + 31) / 32 + (K2 + 31) / 32) * lin16_nh(N) * L16_CHUNK;
}

extern "C" int danbo_linear16_pack_frag(const float* w, long stride_n, long stride_k, int N, int K1, int K2, int frag_in, void* packed,
                                        void* stream) {
    DANBO_CHECK_ARG(w && packed && N >= 1 && N <= L16_MAX_N && K1 >= 1 && K2 >= 0 && frag_in >= 0 && frag_in <= 3);
    DANBO_CHECK_ARG(!(frag_in & 1) || K1 % 32 == 0);
    DANBO_CHECK_ARG(!(frag_in & 2) || (K2 > 0 && K2 % 32 == 0));
    const long total = (long)danbo_linear16_packed_bytes(N, K1, K2) / 2;
    hipLaunchKernelGGL(k_linear16_pack, dim3(stream_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, w, stride_n, stride_k, N, K1,
                       K2, lin16_nh(N), frag_in, 0, (_Float16*)packed);
    DANBO_LAUNCH_RET();
}

/* w [N, 24 (1 + 2 L) + 72 + K2]: a layer whose first 24 (1 + 2 L) + 72 inputs are A-NeRF's density inputs, recomputed by
 * danbo_linear16_fwd_enc from the encoder's table (14 k-steps = DANBO_LINEAR16_ENC_K slots), the other K2 a second input
 * (frag_in bit 1: in fragment order).  packed: danbo_linear16_packed_bytes(N, DANBO_LINEAR16_ENC_K, K2) bytes. */
extern "C" int danbo_linear16_pack_enc(const float* w, long stride_n, long stride_k, int N, int L, int K2, int frag_in, void* packed,
                                       void* stream) {
    DANBO_CHECK_ARG(w && packed && N >= 1 && N <= L16_MAX_N && L >= 1 && L <= 7 && K2 >= 0 && (frag_in == 0 || frag_in == 2));
    DANBO_CHECK_ARG(!(frag_in & 2) || (K2 > 0 && K2 % 32 == 0));
    const long total = (long)danbo_linear16_packed_bytes(N, L16_ENC_K, K2) / 2;
    hipLaunchKernelGGL(k_linear16_pack, dim3(stream_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, w, stride_n, stride_k, N,
                       L16_ENC_K, K2, lin16_nh(N), frag_in, L, (_Float16*)packed);
    DANBO_LAUNCH_RET();
}

extern "C" int danbo_linear16_pack(const float* w, long stride_n, long stride_k, int N, int K1, int K2, void* packed, void* stream) {
    return danbo_linear16_pack_frag(w, stride_n, stride_k, N, K1, K2, 0, packed, stream);
}

extern "C" int danbo_linear16_fwd(const float* x1, int ld1, int K1, const float* x2, int ld2, int K2, const void* packed,
                                  const float* bias, int N, int act, float* y, int ldy, int M, const int32_t* count, void* stream) {
    return danbo_linear16_fwd_frag(x1, ld1, K1, x2, ld2, K2, packed, bias, N, act, y, ldy, M, count, 0, stream);
}

extern "C" int danbo_linear16_fwd_frag(const float* x1, int ld1, int K1, const float* x2, int ld2, int K2, const void* packed,
                                       const float* bias, int N, int act, float* y, int ldy, int M, const int32_t* count, int frag,
                                       void* stream) {
    DANBO_CHECK_ARG(x1 && packed && y && N >= 1 && N <= L16_MAX_N && K1 >= 1 && K2 >= 0 && (K2 == 0 || x2) && M >= 0);
    DANBO_CHECK_ARG((act == 0 || act == 1) && frag >= 0 && frag <= 7);
    DANBO_CHECK_ARG(!(frag & 1) || K1 % 32 == 0);
    DANBO_CHECK_ARG(!(frag & 2) || (K2 > 0 && K2 % 32 == 0));
    DANBO_CHECK_ARG(!(frag & 4) || N % 32 == 0);
    // 16-byte accesses: row strides multiples of 4 floats, rows readable up to a multiple of 4 columns, aligned bases
    DANBO_CHECK_ARG((frag & 1) || (ld1 % 4 == 0 && ld1 >= ((K1 + 3) & ~3)));
    DANBO_CHECK_ARG((frag & 2) || K2 == 0 || (ld2 % 4 == 0 && ld2 >= ((K2 + 3) & ~3)));
    DANBO_CHECK_ARG((frag & 4) || (ldy >= N && ldy % 4 == 0));
    DANBO_CHECK_ARG((uintptr_t)x1 % 16 == 0 && (uintptr_t)x2 % 16 == 0 && (uintptr_t)y % 16 == 0);
    if (M == 0) return 0;
    Lin16Args a{x1, x2, ld1, ld2, K1, K2, (const char*)packed, bias, y, ldy, N, act, M, count, frag, 0, g_lin16_trace};
    const int tiles = (M + L16_BM - 1) / L16_BM;
    const dim3 grid(tiles < num_cu() ? tiles : num_cu()), block(L16_THREADS);
    const int nh = lin16_nh(N), np = ((N + 15) / 16 - 16 * (nh - 1) + 1) / 2;   // tile pairs in the last chunk of a k-step
#define DANBO_L16_GO(NH_, NP_, TR_, FR_)                                                                                   \
    {                                                                                                                      \
        DANBO_ENSURE_LDS((k_linear16<NH_, NP_, TR_, FR_>), L16_LDS_BYTES);                                          \
        hipLaunchKernelGGL((k_linear16<NH_, NP_, TR_, FR_>), grid, block, L16_LDS_BYTES, (hipStream_t)stream, a);   \
    }
#define DANBO_L16_SHAPES(FR_)                                                                                              \
    {                                                                                                                      \
        if (nh == 2 && np == 8) DANBO_L16_GO(2, 8, false, FR_)                                                             \
        else if (nh == 2 && np == 6) DANBO_L16_GO(2, 6, false, FR_) /* N = 448: the A-NeRF trunk */                        \
        else if (nh == 2) DANBO_L16_GO(2, 0, false, FR_)                                                                   \
        else if (np == 8) DANBO_L16_GO(1, 8, false, FR_)            /* N in 225 .. 256 */                                  \
        else DANBO_L16_GO(1, 0, false, FR_)                                                                                \
    }
    if (a.trace && nh == 2 && frag == 0) DANBO_L16_GO(2, 0, true, 0)
    else if (frag == 0) DANBO_L16_SHAPES(0)
    else if (frag == 1 && nh == 1) {             // last layer of a fragment-order trunk: rows out (N <= 256: the wider
        if (np == 8) DANBO_L16_GO(1, 8, false, 1)  // instantiations do not fit the register file, tests/test_host_logic.py)
        else DANBO_L16_GO(1, 0, false, 1)
    }
    else if (frag == 4) DANBO_L16_SHAPES(4)      // first layer: rows in
    else if (frag == 5) DANBO_L16_SHAPES(5)
    else if (frag == 6) DANBO_L16_SHAPES(6)      // skip layer: [rows | fragments] in
    else return DANBO_EINVAL;
#undef DANBO_L16_SHAPES
#undef DANBO_L16_GO
    DANBO_LAUNCH_RET();
}

/* y (fragment order) = act([enc(table) | x2] W^T + bias): the first / the skip layer of the A-NeRF trunk with the 24 (1 + 2 L) + 72
 * density inputs recomputed from the encoder's table [M, 144] (danbo_anerf_encode_compact) instead of read (W packed by
 * danbo_linear16_pack_enc).  x2: NULL, or the second input [M, K2] in fragment order.  N = 448 (the shipped A-NeRF width). */
/* A-NeRF's head layer with the colour head as its epilogue: x1 = the trunk's last activation in fragment order [M, K1]; packed / bias:
 * the (VW + 1)-wide layer [views_linears.0[:, :W] feature_linear ; alpha_linear] (frag_in 1); rows = the samples of rays
 * [ray0, ray0 + M / S) in ray-major order; w [M, 24], C [24, R_total, VW], table, cam_idx, rgb_*: as danbo_anerf_color_fwd.
 * raw_out [R_total, S, 4].  S % 16 == 0, VW % 16 == 0, VW <= 240. */
extern "C" int danbo_linear16_fwd_color(const float* x1, int K1, const void* packed, const float* bias, int VW, int M, const float* w,
                                        const float* C, const float* table, const int64_t* cam_idx, int n_codes, int R_total, int ray0,
                                        int S, const float* rgb_w, const float* rgb_b, float* raw_out, void* stream) {
    DANBO_CHECK_ARG(x1 && packed && bias && w && C && table && rgb_w && rgb_b && raw_out && K1 >= 32 && K1 % 32 == 0 && M >= 0);
    DANBO_CHECK_ARG(VW >= 16 && VW % 16 == 0 && VW <= 240 && S >= 16 && S % 16 == 0 && M % S == 0 && n_codes >= 0 && ray0 >= 0);
    DANBO_CHECK_ARG(ray0 + M / S <= R_total && (uintptr_t)x1 % 16 == 0 && (uintptr_t)w % 16 == 0 && (uintptr_t)table % 16 == 0 &&
                    (uintptr_t)raw_out % 16 == 0);
    if (M == 0) return 0;
    const int N = VW + 1;
    Lin16Args a{x1, nullptr, 0, 0, K1, 0, (const char*)packed, bias, nullptr, 0, N, 0, M, nullptr, 1 | 16, 0, nullptr,
                w, C, table, rgb_w, rgb_b, cam_idx, raw_out, n_codes, R_total, ray0, S, VW};
    const int tiles = (M + L16_BM - 1) / L16_BM;
    const dim3 grid(tiles < num_cu() ? tiles : num_cu()), block(L16_THREADS);
    const int np = ((N + 15) / 16 + 1) / 2;
    if (np == 8) {
        DANBO_ENSURE_LDS((k_linear16<1, 8, false, 17>), L16_COLOR_LDS_BYTES);
        hipLaunchKernelGGL((k_linear16<1, 8, false, 17>), grid, block, L16_COLOR_LDS_BYTES, (hipStream_t)stream, a);
    } else {
        DANBO_ENSURE_LDS((k_linear16<1, 0, false, 17>), L16_COLOR_LDS_BYTES);
        hipLaunchKernelGGL((k_linear16<1, 0, false, 17>), grid, block, L16_COLOR_LDS_BYTES, (hipStream_t)stream, a);
    }
    DANBO_LAUNCH_RET();
}

extern "C" int danbo_linear16_fwd_enc(const float* table, int L, const float* x2, int K2, const void* packed, const float* bias, int N,
                                      int act, float* y, int M, const int32_t* count, void* stream) {
    DANBO_CHECK_ARG(table && packed && y && N == 448 && L >= 1 && L <= 7 && K2 >= 0 && (K2 == 0) == (x2 == nullptr) && M >= 0);
    DANBO_CHECK_ARG((act == 0 || act == 1) && K2 % 32 == 0);
    DANBO_CHECK_ARG((uintptr_t)table % 16 == 0 && (uintptr_t)x2 % 16 == 0 && (uintptr_t)y % 16 == 0);
    if (M == 0) return 0;
    const int frag = 8 | 4 | (x2 ? 2 : 0);
    Lin16Args a{table, x2, L16_ENC_FLOATS, 0, L16_ENC_K, K2, (const char*)packed, bias, y, 0, N, act, M, count, frag, L, nullptr};
    const int tiles = (M + L16_BM - 1) / L16_BM;
    const dim3 grid(tiles < num_cu() ? tiles : num_cu()), block(L16_THREADS);
    if (x2) {
        DANBO_ENSURE_LDS((k_linear16<2, 6, false, 14>), L16_LDS_BYTES);
        hipLaunchKernelGGL((k_linear16<2, 6, false, 14>), grid, block, L16_LDS_BYTES, (hipStream_t)stream, a);
    } else {
        DANBO_ENSURE_LDS((k_linear16<2, 6, false, 12>), L16_LDS_BYTES);
        hipLaunchKernelGGL((k_linear16<2, 6, false, 12>), grid, block, L16_LDS_BYTES, (hipStream_t)stream, a);
    }
    DANBO_LAUNCH_RET();
}
