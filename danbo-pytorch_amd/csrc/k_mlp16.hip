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
#include "common.hpp"

namespace danbo {

#ifndef DANBO_M16_BT
#define DANBO_M16_BT 2   // output tiles per batch of A-fragment reads (2 / 4 / 8 measured within 2 %)
#endif

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int M16_BM = 128;            // rows per workgroup iteration
constexpr int CHUNK_BYTES = 32768;     // 32 fragment pieces of 1 KB
constexpr int RING_SLOTS = 4;
constexpr int NCH_X0 = 7;              // 7 k-steps of 32 PE features (208 >= 195)
constexpr int NCH_ACT = 8;             // 8 k-steps of 32 features
constexpr int NCH_VIEW = 4;            // 8 k-steps, 8 output tiles -> 2 k-steps per chunk
// feature_linear (256->256, no activation) and the per-sample part of views_linears.0 (256->128) are
// merged into ONE 256->128 GEMM at pack time: W_fv = W_v[:, :256] W_f, bias W_v[:, :256] b_f folded into
// the per-ray view constants (65 536 fewer MACs per row than the reference's 677 376)
constexpr int NCH_TOTAL = NCH_X0 + 4 * NCH_ACT + (NCH_X0 + NCH_ACT) + 2 * NCH_ACT + NCH_VIEW;  // 74
static_assert(NCH_TOTAL * CHUNK_BYTES == DANBO_MLP16_PACKED_BYTES, "header constant out of date");
constexpr int X0_KSTEPS = 7;
constexpr int IN_CH = 195, W_ = 256, VW_ = 128;

// ---------------------------------------------------------------------------------------------
// weight packing: one thread per half element
// ---------------------------------------------------------------------------------------------
struct Pack16Args {
    const float* pts_w[8];
    const float* feature_w;
    const float* feature_b;
    const float* views_w;
    const float* views_b;
    int Cv;
    float* views_b_eff;  // [128] = views_b + W_v[:, :256] feature_b
};

__global__ __launch_bounds__(256) void k_mlp16_pack(Pack16Args a, _Float16* __restrict__ packed) {
    const long total = (long)NCH_TOTAL * (CHUNK_BYTES / 2);
    if (blockIdx.x == 0 && threadIdx.x < VW_) {
        const int n = threadIdx.x;
        double acc = 0.0;
        for (int c = 0; c < W_; ++c) acc += (double)a.views_w[(size_t)n * (W_ + a.Cv) + c] * (double)a.feature_b[c];
        a.views_b_eff[n] = (float)((double)a.views_b[n] + acc);
    }
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int chunk = (int)(idx / (CHUNK_BYTES / 2));
        const int within = (int)(idx % (CHUNK_BYTES / 2));
        const int piece = within >> 9, lane = (within >> 3) & 63, e = within & 7;
        const int q = lane >> 4;
        // which GEMM / which part does this chunk belong to
        int layer, cl, kind;  // kind 0: x0 part, 1: act part, 2: view layer
        if (chunk < 7) { layer = 0; cl = chunk; kind = 0; }
        else if (chunk < 39) { layer = 1 + (chunk - 7) / 8; cl = (chunk - 7) % 8; kind = 1; }
        else if (chunk < 46) { layer = 5; cl = chunk - 39; kind = 0; }
        else if (chunk < 54) { layer = 5; cl = chunk - 46; kind = 1; }
        else if (chunk < 70) { layer = 6 + (chunk - 54) / 8; cl = (chunk - 54) % 8; kind = 1; }
        else { layer = 9; cl = chunk - 70; kind = 2; }
        int s, T, hl;
        if (kind == 2) { s = 2 * cl + (piece >> 4); T = (piece >> 1) & 7; hl = piece & 1; }
        else { s = cl; T = piece >> 1; hl = piece & 1; }
        const int n = 16 * T + (lane & 15);
        float w = 0.f;
        if (kind == 0) {
            const int j = 8 * s + e;
            const int c = j / 13, t = j % 13, kk = q + 4 * c;
            if (j < 52 && kk < FEAT) {
                const int col = FEAT * t + kk;  // [x | sin 2^0 | cos 2^0 | ...] blocks of 15
                w = layer == 0 ? a.pts_w[0][(size_t)n * IN_CH + col] : a.pts_w[5][(size_t)n * (IN_CH + W_) + col];
            }
        } else {
            const int f = 16 * (2 * s + (e >> 2)) + 4 * q + (e & 3);
            if (layer == 5) w = a.pts_w[5][(size_t)n * (IN_CH + W_) + IN_CH + f];
            else if (layer <= 7) w = a.pts_w[layer][(size_t)n * W_ + f];
            else {  // merged feature + view layer: W_fv[n][f] = sum_c W_v[n][c] W_f[c][f]
                double acc = 0.0;
                const float* wv = a.views_w + (size_t)n * (W_ + a.Cv);
                for (int c = 0; c < W_; ++c) acc += (double)wv[c] * (double)a.feature_w[(size_t)c * W_ + f];
                w = (float)acc;
            }
        }
        const _Float16 hi = (_Float16)w;
        packed[idx] = hl ? (_Float16)(w - (float)hi) : hi;
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

constexpr int M16_THREADS = 512;
constexpr int M16_TABLE_FLOATS = 8 * W_ + W_ + 3 * VW_ + 4;
constexpr int M16_LDS_BYTES = RING_SLOTS * CHUNK_BYTES + M16_TABLE_FLOATS * 4 + 8 * (1024 + 256);  // + staging, see TileSrc

struct Pipe {
    const char* packed;
    char* ring;
    int issue_chunk, issue_slot, cons_slot, wave, lane;
    bool early;
};

// Tile-boundary prefetch, so that no wavefront waits on HBM between two row tiles:
//   * the NEXT tile's rows (blended features h and list entries) are fetched by two LDS-DMA loads per wavefront
//     into a 1.25 KB staging area while the view layer of the current tile runs;
//   * the CURRENT tile's per-ray view constants are fetched by eight inline-asm loads one chunk before the
//     colour head needs them (inline asm: a compiler-tracked load would make the compiler wait with
//     s_waitcnt vmcnt(0), which also drains the weight ring).
// Both ride on the ring's in-order vmcnt accounting: see pipe_handover.
constexpr int STAGE_H_BYTES = 1024, STAGE_BYTES = 1024 + 256;  // per wavefront: h [16][16] floats, list [64] ints
struct TileSrc {
    const float* h;        // a.h
    const int32_t* list;   // a.list or nullptr
    const char* dummy;     // any readable 256 bytes (the packed weights)
    int next_row0;         // first row of this wavefront in the next tile
    int n;
    char* stage;           // this wavefront's staging area (wave-uniform)
};

__device__ __forceinline__ void prefetch_rows(const TileSrc& t, int lane) {
    int row0 = t.next_row0;
    asm volatile("" : "+s"(row0));  // addresses are formed here, not hoisted out of the layer loop and spilled
    const int rh = min(row0 + (lane >> 2), t.n - 1);
    const float* src_h = t.h + (size_t)rh * DANBO_H_STRIDE + 4 * (lane & 3);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src_h,
                                     (__attribute__((address_space(3))) void*)t.stage, 16, 0, 0);
    const int rl = min(row0 + (lane & 15), t.n - 1);
    const void* src_l = t.list ? (const void*)(t.list + rl) : (const void*)(t.dummy + 4 * lane);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src_l,
                                     (__attribute__((address_space(3))) void*)(t.stage + STAGE_H_BYTES), 4, 0, 0);
}
// one base address + instruction offsets: no per-load address registers.  Four column tiles per call.
template <int HALF>
__device__ __forceinline__ void prefetch_cv(const float* base, f32x4 (&cv)[8]) {
#define DANBO_CV_LOAD(T) asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(cv[T]) : "v"(base), "i"((T) * 64) : "memory")
    DANBO_CV_LOAD(4 * HALF); DANBO_CV_LOAD(4 * HALF + 1); DANBO_CV_LOAD(4 * HALF + 2); DANBO_CV_LOAD(4 * HALF + 3);
#undef DANBO_CV_LOAD
}

// every wavefront loads 4 of the 32 pieces of a chunk
__device__ __forceinline__ void pipe_issue(Pipe& p) {
    const char* src = p.packed + (size_t)p.issue_chunk * CHUNK_BYTES + p.wave * 4096 + p.lane * 16;
    char* dst = p.ring + p.issue_slot * CHUNK_BYTES + p.wave * 4096;
#pragma unroll
    for (int q = 0; q < 4; ++q)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + q * 1024),
                                         (__attribute__((address_space(3))) void*)(dst + q * 1024), 16, 0, 0);
    p.issue_chunk = p.issue_chunk + 1 == NCH_TOTAL ? 0 : p.issue_chunk + 1;
    p.issue_slot = p.issue_slot + 1 == RING_SLOTS ? 0 : p.issue_slot + 1;
}

// Ring hand-over #c, executed once per chunk c by every wavefront -- by the "early" wavefronts (0-3) in the
// middle of chunk c, by the "late" ones (4-7, their SIMD partners) before they start it, so the two
// wavefronts of a SIMD run half a chunk apart and one's VALU epilogue work and LDS latencies fall under
// the other's MFMAs instead of both stalling at the same program point:
//   wait: my share of chunk c+1 has landed (<= 4 younger loads = chunk c+2 outstanding);
//   barrier: everybody's has, and everybody is past chunk c-1;  then refill that slot with chunk c+3.
// WAIT / extra (view layer only): `extra()` issues additional loads between the barrier and the ring refill, so
// they are OLDER than that refill and YOUNGER than the chunk the next hand-over waits for; that next hand-over
// therefore allows WAIT = 4 + (number of extra loads) operations to stay in flight.
struct NoExtra {
    __device__ __forceinline__ void operator()() const {}
};
template <int WAIT, class Extra>
__device__ __forceinline__ void pipe_handover(Pipe& p, const Extra& extra) {
    static_assert(WAIT == 4 || WAIT == 6 || WAIT == 8, "add the s_waitcnt immediate");
    if (WAIT == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if (WAIT == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    extra();
    pipe_issue(p);
}

__device__ __forceinline__ half8 lds_frag(const char* base, int piece) {
    return *reinterpret_cast<const half8*>(base + piece * 1024);
}

// hi*hi + hi*lo + lo*hi into acc; FIRST: the accumulator starts from zero (no separate clear)
template <bool FIRST>
__device__ __forceinline__ void mfma3(f32x4& acc, const half8& ah, const half8& al, const half8& bh, const half8& bl) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, FIRST ? f32x4{0.f, 0.f, 0.f, 0.f} : acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc, 0, 0, 0);
}

// One 32 KB chunk = 16 (tile, hi/lo) fragment pairs.  DENSE layers: one k-step, output tiles 0..15, B = (b0h, b0l).
// VIEW layer: two k-steps of 8 output tiles, B = b0 for pairs 0..7 and b1 for pairs 8..15.
template <int NACC, bool VIEW, bool FIRST, int WAIT = 4, class Extra = NoExtra>
__device__ __forceinline__ void chunk_mfma(f32x4 (&acc)[NACC], Pipe& p, const half8& b0h, const half8& b0l,
                                           const half8& b1h, const half8& b1l, const Extra& extra = Extra()) {
    if (!p.early) pipe_handover<WAIT>(p, extra);
    const char* base = p.ring + p.cons_slot * CHUNK_BYTES + p.lane * 16;
    p.cons_slot = p.cons_slot + 1 == RING_SLOTS ? 0 : p.cons_slot + 1;
    // batches of BT tiles: 2 BT ds_read_b128, then their 3 BT MFMAs.  (Reading a batch ahead buys nothing with
    // compiler-tracked LDS loads -- the compiler waits with lgkmcnt(0), i.e. for the look-ahead batch too; the other
    // wavefront of the SIMD covers the read latency.)
    constexpr int BT = DANBO_M16_BT;
#pragma unroll
    for (int b = 0; b < 16 / BT; ++b) {
        half8 ah[BT], al[BT];
#pragma unroll
        for (int t = 0; t < BT; ++t) {
            ah[t] = lds_frag(base, 2 * (BT * b + t));
            al[t] = lds_frag(base, 2 * (BT * b + t) + 1);
        }
        if (BT * b == 8 && p.early) pipe_handover<WAIT>(p, extra);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < BT; ++t) {
            const int T = BT * b + t;
            if (VIEW && T >= 8) mfma3<false>(acc[T - 8], ah[t], al[t], b1h, b1l);
            else mfma3<FIRST>(acc[T], ah[t], al[t], b0h, b0l);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

__device__ __forceinline__ void split8(const float* v, half8& hi, half8& lo) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const _Float16 hh = (_Float16)v[e];
        hi[e] = hh;
        lo[e] = (_Float16)(v[e] - (float)hh);
    }
}

// sum of the four lane-group partials of a sample; identical in all four groups
__device__ __forceinline__ float quad_sum(float p) {
    p += lane_xor16(p);
    p += lane_xor32(p);
    return p;
}

// B fragments of k-step s of the next GEMM from the previous layer's accumulators: tiles 2s and 2s+1,
// bias + ReLU, hi/lo split.  ALPHA: also accumulate this lane's part of the density logit.
template <bool ALPHA>
__device__ __forceinline__ void act_fragment(const f32x4& a0, const f32x4& a1, const float* bias /* + 4qq + 32s */,
                                             const float* aw, float& alpha_part, half8& bh, half8& bl) {
    float v[8];
    const float4 b0 = *reinterpret_cast<const float4*>(bias);
    const float4 b1 = *reinterpret_cast<const float4*>(bias + 16);
    v[0] = fmaxf(a0[0] + b0.x, 0.f); v[1] = fmaxf(a0[1] + b0.y, 0.f);
    v[2] = fmaxf(a0[2] + b0.z, 0.f); v[3] = fmaxf(a0[3] + b0.w, 0.f);
    v[4] = fmaxf(a1[0] + b1.x, 0.f); v[5] = fmaxf(a1[1] + b1.y, 0.f);
    v[6] = fmaxf(a1[2] + b1.z, 0.f); v[7] = fmaxf(a1[3] + b1.w, 0.f);
    if (ALPHA) {
        const float4 w0 = *reinterpret_cast<const float4*>(aw);
        const float4 w1 = *reinterpret_cast<const float4*>(aw + 16);
        alpha_part = fmaf(v[0], w0.x, alpha_part); alpha_part = fmaf(v[1], w0.y, alpha_part);
        alpha_part = fmaf(v[2], w0.z, alpha_part); alpha_part = fmaf(v[3], w0.w, alpha_part);
        alpha_part = fmaf(v[4], w1.x, alpha_part); alpha_part = fmaf(v[5], w1.y, alpha_part);
        alpha_part = fmaf(v[6], w1.z, alpha_part); alpha_part = fmaf(v[7], w1.w, alpha_part);
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

__global__ __launch_bounds__(M16_THREADS, 2) void k_pe_mlp16(Mlp16Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* s_bias = reinterpret_cast<float*>(smem + RING_SLOTS * CHUNK_BYTES);  // [8][256]
    float* s_aw = s_bias + 8 * W_;                                              // [256]
    float* s_rgbw = s_aw + W_;                                                  // [3][128]
    float* s_misc = s_rgbw + 3 * VW_;                                           // alpha_b, rgb_b[3]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 15, qq = lane >> 4;
    for (int i = tid; i < 8 * W_; i += M16_THREADS) s_bias[i] = a.pts_b[i >> 8][i & 255];
    if (tid < W_) s_aw[tid] = a.alpha_w[tid];
    for (int i = tid; i < 3 * VW_; i += M16_THREADS) s_rgbw[i] = a.rgb_w[i];
    if (tid < 4) s_misc[tid] = tid == 0 ? a.alpha_b[0] : a.rgb_b[tid - 1];

    const int n = resolve_count(a.count, a.n_cap);
    const int ntiles = (n + M16_BM - 1) / M16_BM;
    if ((int)blockIdx.x >= ntiles) return;

    Pipe p;
    p.packed = a.packed; p.ring = smem; p.issue_chunk = 0; p.issue_slot = 0; p.cons_slot = 0; p.wave = wave; p.lane = lane;
    p.early = wave < 4;  // wavefronts w and w+4 of a workgroup share a SIMD
    pipe_issue(p);
    pipe_issue(p);
    pipe_issue(p);
    // first tile of this workgroup: the same two staging loads (later tiles: issued during the previous view layer)
    TileSrc src;
    src.h = a.h; src.list = a.list; src.dummy = a.packed; src.n = n;
    src.stage = smem + RING_SLOTS * CHUNK_BYTES + M16_TABLE_FLOATS * 4 + wave * STAGE_BYTES;
    src.next_row0 = blockIdx.x * M16_BM + wave * 16;
    prefetch_rows(src, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // ring chunks 0-2 and the first rows (tables: the same barrier)
    __syncthreads();

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        // ------------------------------------------------------------------ inputs (staged during the previous tile)
        const int row = tile * M16_BM + wave * 16 + m;
        const bool row_ok = row < n;
        const float* sh = reinterpret_cast<const float*>(src.stage) + m * DANBO_H_STRIDE + qq;
        const int staged_dst = reinterpret_cast<const int*>(src.stage + STAGE_H_BYTES)[m];
        int dst = row_ok ? (a.list ? staged_dst : row) : -1;
        asm volatile("" : "+v"(dst));  // materialised now: the staging area is overwritten during this tile's view layer
        const int ray = dst >= 0 ? dst / a.S : 0;
        // this lane's 4 channels: kk = qq + 4c  (kk = 15 is padding)
        float hv[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) hv[c] = row_ok ? sh[4 * c] : 0.f;
        if (qq == 3) hv[3] = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) asm volatile("" : "+v"(hv[c]));
        float alpha_part = 0.f;
        f32x4 prev[16];  // pre-bias outputs of the previous layer
        src.next_row0 = (tile + (int)gridDim.x) * M16_BM + wave * 16;
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
#define DANBO_X0_STEP(KS, FIRST_)                                                  \
                {                                                                  \
                    float v8[8], hk[4] = {hv[0], hv[1], hv[2], hv[3]};             \
                    /* re-defined after the previous chunk: the sincos of later k-steps must not be hoisted (and spilled) */ \
                    asm volatile("" : "+v"(hk[0]), "+v"(hk[1]), "+v"(hk[2]), "+v"(hk[3]));                            \
                    pe_kstep<KS>(hk, cs_keep, v8);                                 \
                    half8 xh, xl;                                                  \
                    split8(v8, xh, xl);                                            \
                    chunk_mfma<16, false, FIRST_>(acc, p, xh, xl, xh, xl);         \
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
            if (step != 0 && step != 8) {
                const float* bias = s_bias + (step - 1) * W_ + q4;
                {
                    half8 bh, bl;
                    act_fragment<false>(prev[0], prev[1], bias, nullptr, alpha_part, bh, bl);
                    if (step == 5) chunk_mfma<16, false, false>(acc, p, bh, bl, bh, bl);
                    else chunk_mfma<16, false, true>(acc, p, bh, bl, bh, bl);
                }
#pragma unroll
                for (int s = 1; s < NCH_ACT; ++s) {
                    half8 bh, bl;
                    int off_s = 32 * s;
                    asm volatile("" : "+v"(off_s));   // no hoisting of all eight bias loads (the offset, not the pointer:
                    const float* bias_s = bias + off_s;  // the pointer must stay an LDS pointer)
                    act_fragment<false>(prev[2 * s], prev[2 * s + 1], bias_s, nullptr, alpha_part, bh, bl);
                    chunk_mfma<16, false, false>(acc, p, bh, bl, bh, bl);
                }
            }
            if (step == 8) {
                // view layer; while it runs: stage the next tile's rows (chunk 0)
                const float* bias = s_bias + 7 * W_ + q4;
                const float* aw = s_aw + q4;
                f32x4 (&accv)[8] = *reinterpret_cast<f32x4 (*)[8]>(&acc[0]);
#pragma unroll
                for (int c = 0; c < NCH_VIEW; ++c) {
                    half8 b0h, b0l, b1h, b1l;
                    int off_c = 64 * c;
                    asm volatile("" : "+v"(off_c));
                    const float* bias_c = bias + off_c;
                    const float* aw_c = aw + off_c;
                    act_fragment<true>(prev[4 * c], prev[4 * c + 1], bias_c, aw_c, alpha_part, b0h, b0l);
                    act_fragment<true>(prev[4 * c + 2], prev[4 * c + 3], bias_c + 32, aw_c + 32, alpha_part, b1h, b1l);
                    if (c == 0) chunk_mfma<8, true, true, 4>(accv, p, b0h, b0l, b1h, b1l, [&]() { prefetch_rows(src, lane); });
                    else if (c == 1) chunk_mfma<8, true, false, 6>(accv, p, b0h, b0l, b1h, b1l);
                    else chunk_mfma<8, true, false, 4>(accv, p, b0h, b0l, b1h, b1l);
                }
#pragma unroll
                for (int T = 8; T < 16; ++T) acc[T] = acc[T - 8];  // defined values for the copy below (never read)
            }
#pragma unroll
            for (int T = 0; T < 16; ++T) prev[T] = acc[T];
        }
        f32x4 (&accv)[8] = *reinterpret_cast<f32x4 (*)[8]>(&prev[0]);
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
#pragma unroll
        for (int T = 0; T < 8; ++T) {
            const int nn = 16 * T;  // + 4*qq + i
            f32x4 c4 = cvq[T];
            if (!a.cview) c4 = f32x4{0.f, 0.f, 0.f, 0.f};
            const float pre[4] = {accv[T][0], accv[T][1], accv[T][2], accv[T][3]};
            if (aux) *reinterpret_cast<float4*>(aux + nn) = make_float4(pre[0], pre[1], pre[2], pre[3]);
            const float x[4] = {fmaxf(pre[0] + c4[0], 0.f), fmaxf(pre[1] + c4[1], 0.f), fmaxf(pre[2] + c4[2], 0.f),
                                fmaxf(pre[3] + c4[3], 0.f)};
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
        if (qq == 0 && dst >= 0) {
            reinterpret_cast<float4*>(a.raw_out)[dst] = make_float4(r_, g_, b_, al);
            if (a.aux_out) a.aux_out[(size_t)row * (VW_ + 1) + VW_] = al;
        }
    }
    // every wavefront executed the same number of hand-overs; drain the ring before the LDS is released
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}

}  // namespace danbo

using namespace danbo;

extern "C" int danbo_mlp16_pack(const float* const* pts_w, const float* feature_w, const float* feature_b,
                                 const float* views_w, const float* views_b, int Cv, void* packed16,
                                 float* views_b_eff, void* stream) {
    DANBO_CHECK_ARG(pts_w && feature_w && feature_b && views_w && views_b && packed16 && views_b_eff && Cv >= 0);
    Pack16Args a;
    for (int i = 0; i < 8; ++i) a.pts_w[i] = pts_w[i];
    a.feature_w = feature_w;
    a.feature_b = feature_b;
    a.views_w = views_w;
    a.views_b = views_b;
    a.Cv = Cv;
    a.views_b_eff = views_b_eff;
    hipLaunchKernelGGL(k_mlp16_pack, dim3(2048), dim3(256), 0, (hipStream_t)stream, a, reinterpret_cast<_Float16*>(packed16));
    DANBO_LAUNCH_RET();
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
    const int ntiles = ceil_div(n, M16_BM);
    const int grid = ntiles < num_cu() ? ntiles : num_cu();
    hipLaunchKernelGGL(k_pe_mlp16, dim3(grid), dim3(M16_THREADS), M16_LDS_BYTES, (hipStream_t)stream, a);
    DANBO_LAUNCH_RET();
}
