// K1b + K2 (fast variant): per-sample factorised gather, bone-assignment GNN, masked sigmoid and
// blend with the two per-bone GEMMs on the half-precision matrix cores (fp16 hi/lo split operands,
// three MFMAs per fp32-accurate product, fp32 accumulation -- same scheme as csrc/k_mlp16.hip).
// gfx950 only.
//
// One wavefront = 32 samples, lane = sample + 32*half.  Transposed formulation per bone j:
//   z0_j^T [32 x samples] = sum_{j' in N(j)+{j}} (adjw[j][j'] W0_j'^T) [32 x 16] * f_j'^T [16 x samples]
//   z1_j^T [32 x samples] = W1_j^T [32 x 32] * relu(z0_j + b0)^T
// i.e. the 70-nonzero skeleton adjacency is folded into the layer-0 weights at pack time, and
// the output tile of layer 0 (lane (m,h), reg r <-> channel 8(r>>2)+4h+(r&3)) is directly the B
// fragment of layer 1.  Each lane computes 8 of the 15 gathered features of every bone for its
// sample (h=0: features 0..7, h=1: 8..14) as fp16 hi/lo fragments.
// Weights (236 KB in fragment order, L2-resident): the 8 .. 14 one-KB pieces of a bone are loaded into registers by the
// wavefront that evaluates it, and kept while the bone stays the same (see k_assign16 below).
#include "common.hpp"

namespace danbo {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int A16_CHUNK = 32768;
constexpr int A16_NCHUNK = 8;  // 236 pieces of 1 KB, padded to 256
static_assert(A16_NCHUNK * A16_CHUNK == DANBO_ASSIGN16_PACKED_BYTES, "header constant out of date");
constexpr int A16_BM = 128;    // samples per workgroup iteration (4 wavefronts x 32)

// SMPL tree neighbours (parent + children), the bone itself first
__host__ __device__ constexpr int a16_deg(int j) {
    constexpr int d[J] = {3, 2, 2, 2, 2, 2, 2, 2, 2, 4, 1, 1, 2, 2, 2, 1, 2, 2, 2, 2, 2, 2, 1, 1};
    return d[j];
}
__host__ __device__ constexpr int a16_nb(int j, int q) {  // q = 0: self
    constexpr int nb[J][5] = {
        {0, 1, 2, 3, -1},    {1, 0, 4, -1, -1},   {2, 0, 5, -1, -1},   {3, 0, 6, -1, -1},   {4, 1, 7, -1, -1},
        {5, 2, 8, -1, -1},   {6, 3, 9, -1, -1},   {7, 4, 10, -1, -1},  {8, 5, 11, -1, -1},  {9, 6, 12, 13, 14},
        {10, 7, -1, -1, -1}, {11, 8, -1, -1, -1}, {12, 9, 15, -1, -1}, {13, 9, 16, -1, -1}, {14, 9, 17, -1, -1},
        {15, 12, -1, -1, -1}, {16, 13, 18, -1, -1}, {17, 14, 19, -1, -1}, {18, 16, 20, -1, -1}, {19, 17, 21, -1, -1},
        {20, 18, 22, -1, -1}, {21, 19, 23, -1, -1}, {22, 20, -1, -1, -1}, {23, 21, -1, -1, -1}};
    return nb[j][q];
}
// first 1-KB piece of bone j in the packed stream: per bone 2*(deg+1) layer-0 pieces + 4 layer-1 pieces
__host__ __device__ constexpr int a16_piece0(int j) {
    int p = 0;
    for (int i = 0; i < j; ++i) p += 2 * (a16_deg(i) + 1) + 4;
    return p;
}
static_assert(a16_piece0(J) == 236, "piece count");

// ---------------------------------------------------------------------------------------------
struct A16PackArgs {
    const float* w0;    // [24][15][32]
    const float* adjw;  // [24][24]
    const float* w1;    // [24][32][32]
};

__global__ __launch_bounds__(256) void k_assign16_pack(A16PackArgs a, _Float16* __restrict__ packed) {
    const int total = A16_NCHUNK * (A16_CHUNK / 2);
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int piece = idx >> 9, lane = (idx >> 3) & 63, e = idx & 7;
        const int n = lane & 31, h = lane >> 5;
        float w = 0.f;
        int hl = 0;
        if (piece < 236) {
            int j = 0;
            while (j + 1 < J && a16_piece0(j + 1) <= piece) ++j;
            const int local = piece - a16_piece0(j);
            const int n0 = 2 * (a16_deg(j) + 1);
            hl = local & 1;
            if (local < n0) {  // layer 0, neighbour q
                const int jp = a16_nb(j, local >> 1);
                const int k = 8 * h + e;
                if (k < FEAT) w = a.adjw[j * J + jp] * a.w0[((size_t)jp * FEAT + k) * 32 + n];
            } else {  // layer 1, k-step ks
                const int ks = (local - n0) >> 1;
                const int c = 8 * (2 * ks + (e >> 2)) + 4 * h + (e & 3);
                w = a.w1[((size_t)j * 32 + c) * 32 + n];
            }
        }
        const _Float16 hi = (_Float16)w;
        packed[idx] = hl ? (_Float16)(w - (float)hi) : hi;
    }
}

// ---------------------------------------------------------------------------------------------
struct A16Args {
    const float* rays_o;
    const float* rays_d;
    const float* z;
    const float* pts;
    int R, S, G;
    const float* skts;
    const float* align;
    const float* axis_scale;
    const float* volumes;
    const uint32_t* valid_bits;
    const int32_t* list;
    const int32_t* count;
    int n_cap;
    const char* packed;
    const float* b0;  // [32]
    const float* b1;  // [24][32]
    const float* w2;  // [24][32]
    const float* b2;  // [24]
    float* h_out;     // [n][16]
    float* confd;     // [n][24] or NULL
    // TRAIN instantiation only (danbo_gather_assign_blend16_train):
    const int32_t* first;   // device scalar: rows [*first, count) of list / h_out / confd are processed, or nullptr (0)
    long long* trace;       // dev tool (tools/micro_assign.py --trace): s_memtime stamps of one wavefront, or nullptr
    unsigned* ticket;       // device word, 0 at launch, 0 again when the launch has finished: the next 128-row tile to hand out
    int no_skip;            // dev (DANBO_A16_NOSKIP=1): evaluate every neighbour pair
};

// per bone: neighbours (self first), number of layer-0 terms, first 1-KB piece of the packed stream
constexpr int A16_NBI = 8;        // ints per bone: nb[0..4], nq = deg + 1, piece0, pad
constexpr int A16_TABLE_FLOATS = 32 + J * 32 + J * 32 + J + J * 16 + J * 4 + 8 /*pad*/ + J * VOL + J * 16 + J * A16_NBI + J * 4 + J;
constexpr int A16_LDS_BYTES = (A16_TABLE_FLOATS * 4 + 15) & ~15;
constexpr int A16_TICKETS_AHEAD = 4;   // draws of a workgroup that come back without a tile: the one that ends it + three in its pipeline

// One bone's 15 windowed features for K2's matrix-core operands: the same quantities as sample_math.hpp's bone_local +
// gather_bone_features (reference core/encoders.py:288-303,442-444, gnn_backbone.py:802-826, misc.py:331-351) in ~320 instead of
// ~400 instructions -- the feature phase is VALU-issue bound (s_memtime trace, kernel comment below).
// What may NOT change is the rounding sequence point -> x -> interpolation weight: the reference forms the cell coordinate as
// ((x + 1) res - 1) / 2 in fp32, whose first addition quantises x to 1.2e-7 ABSOLUTE, the weight inherits 1e-6, and the blended
// feature goes through sin(32 h) and the MLP -- a mathematically better x (fused transforms, one fma for the coordinate) moved the
// raw logits by 1e-4 of their range, the whole north_star budget.  So: the two rigid transforms unfused in the reference's order,
// x = p / |scale| correctly rounded (Markstein: q = p r, q + fma(-q, s, p) r with r = RN(1 / s) from the table), the coordinate in
// the reference's order, exp by expf.  What does change, at the 1e-7 level: the window and the zero padding are folded into the two
// interpolation weights of each axis (one multiply + one fma per feature, unconditional clamped reads).
// The LDS (staged pose) and the global (another pose inside the tile) route run THIS code: a row's result does not depend on
// which one it takes.  abs_scale: [4] = |scale_x|, |scale_y|, |scale_z|, -; inv_scale likewise.
// (in two parts: the coordinates x and the window -- 80 of the instructions -- and the gather; K2 looks at the window of a
// neighbour bone before it pays for that bone's gather)
__device__ __forceinline__ float a16_bone_coords(const float* __restrict__ skt, const float* __restrict__ align,
                                                 const float* __restrict__ abs_scale, const float* __restrict__ inv_scale,
                                                 const float* pnt, float* x) {
    float pt[3];
    bone_local(skt, align, pnt, pt);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float q = mul_rn(pt[k], inv_scale[k]);
        x[k] = fmaf(fmaf(-q, abs_scale[k], pt[k]), inv_scale[k], q);
    }
    return coord_window(x);
}

template <typename VolPtr>
__device__ __forceinline__ void a16_bone_gather(VolPtr vol, const float* x, float win, float* out) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float iy = mul_rn(sub_rn(mul_rn(add_rn(x[k], 1.0f), (float)VRES), 1.0f), 0.5f);
        const float fl = floorf(iy);
        const float w1 = sub_rn(iy, fl);
        // clamp before the int conversion: far-away samples have |iy| ~ 1e3..1e6
        const int y0 = (int)fminf(fmaxf(fl, -2.0f), (float)VRES + 1.0f);
        const int y1 = y0 + 1;
        const bool ok0 = (unsigned)y0 < (unsigned)VRES, ok1 = (unsigned)y1 < (unsigned)VRES;
        const float a0 = ok0 ? mul_rn(sub_rn(1.0f, w1), win) : 0.f, a1 = ok1 ? mul_rn(w1, win) : 0.f;
        const int c0 = (ok0 ? y0 : 0) * 3 + k, c1 = (ok1 ? y1 : 0) * 3 + k;
#pragma unroll
        for (int f = 0; f < VOXF; ++f) out[f * 3 + k] = fmaf(vol[f * (VRES * 3) + c1], a1, mul_rn(vol[f * (VRES * 3) + c0], a0));
    }
}

template <typename VolPtr>
__device__ __forceinline__ void a16_bone_features(const float* __restrict__ skt, const float* __restrict__ align, VolPtr vol,
                                                  const float* __restrict__ abs_scale, const float* __restrict__ inv_scale,
                                                  const float* pnt, float* out) {
    float x[3];
    const float win = a16_bone_coords(skt, align, abs_scale, inv_scale, pnt, x);
    a16_bone_gather(vol, x, win, out);
}

// three instructions per pair: common.hpp.  They are inline asm, which the compiler's hazard recognizer does not see as VALU writes:
// where an MFMA reads the fragments right behind the split (layer 0 since round 5) the wait states are written out, tied to the
// two registers so that the statement stays between the split and the MFMAs.
__device__ __forceinline__ void a16_split8(const float* v, half8& hi, half8& lo) {
    split8_mix(&v[0], hi, lo);
    asm volatile("s_nop 1" : "+v"(hi), "+v"(lo));
}

// Round 4 structure.  A wavefront = 32 rows (lane = row + 32 * half) and loops, at RUN time, over the bones valid for at least
// one of them (grouped rows: 1.7 on the bench frame).  Per bone j: the features of j and its tree neighbours are gathered for
// the wavefront's samples (lane half h takes neighbour 2t + h, the halves then trade 8 values: the B-fragment layout), the
// 2 (deg + 1) + 4 weight fragments of the bone sit in registers -- loaded from L2 only when the bone differs from the one the
// wavefront evaluated last -- and the three layers, the masked sigmoid and the blend follow as before.
// Rounds 1-3 unrolled the bone loop 24 x at compile time (features of all 24 bones in 192 registers, 495 VGPRs = one wavefront
// per SIMD, 27 000 instructions = three times the instruction cache) and streamed all 236 KB of weights through an LDS ring per
// 128-row tile.  An s_memtime trace (tools/micro_assign.py --trace) put 55-70 % of a tile into the feature phase at ~13 cycles
// per instruction: latency nothing covered.  Now: ~2 500 instructions, <= 256 VGPRs and 34 KB of LDS = two wavefronts per SIMD.
// TRAIN: device-side first row, and the pad slot h[15] receives q = sum_j p_j valid_j (the row's assignment mass, which the
// soft-softmax loss compares with its label -- reference core/trainer.py:507-536)
template <bool TRAIN>
__global__ __launch_bounds__(256, 2) void k_assign16(A16Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* s_b0 = reinterpret_cast<float*>(smem);                          // [32]
    float* s_b1 = s_b0 + 32;                                               // [24][32]
    float* s_w2 = s_b1 + J * 32;                                           // [24][32]
    float* s_b2 = s_w2 + J * 32;                                           // [24]
    float* s_align = s_b2 + J;                                             // [24][16]
    float* s_scale = s_align + J * 16;                                     // [24][4]
    float* s_vol = s_scale + J * 4 + 8;                                    // [24][240] volumes of the staged pose (16-B aligned)
    float* s_skt = s_vol + J * VOL;                                        // [24][16]  its bone transforms
    int* s_nbi = reinterpret_cast<int*>(s_skt + J * 16);                   // [24][8]
    float* s_inv = reinterpret_cast<float*>(s_nbi + J * A16_NBI);          // [24][4] RN(1 / |scale|)
    float* s_vmax = s_inv + J * 4;                                         // [24] largest |entry| of the staged pose's volume of each bone

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = lane & 31, hh = lane >> 5;
    if (tid < 32) s_b0[tid] = a.b0[tid];
    for (int i = tid; i < J * 32; i += 256) { s_b1[i] = a.b1[i]; s_w2[i] = a.w2[i]; }
    if (tid < J) s_b2[tid] = a.b2[tid];
    for (int i = tid; i < J * 16; i += 256) s_align[i] = a.align[i];
    for (int i = tid; i < J * 4; i += 256) {
        s_scale[i] = (i & 3) < 3 ? fabsf(a.axis_scale[(i >> 2) * 3 + (i & 3)]) : 1.f;
        s_inv[i] = div_rn(1.0f, s_scale[i]);
    }
    if (tid < J) {
#pragma unroll
        for (int jj = 0; jj < J; ++jj)
            if (jj == tid) {
#pragma unroll
                for (int q = 0; q < 5; ++q) s_nbi[jj * A16_NBI + q] = a16_nb(jj, q) < 0 ? jj : a16_nb(jj, q);
                s_nbi[jj * A16_NBI + 5] = a16_deg(jj) + 1;
                s_nbi[jj * A16_NBI + 6] = a16_piece0(jj);
                s_nbi[jj * A16_NBI + 7] = 0;
            }
    }
    __syncthreads();

    int n = resolve_count(a.count, a.n_cap);
    if (TRAIN && a.first != nullptr) {
        const int f0 = *a.first;
        n = n > f0 ? n - f0 : 0;
        a.list += f0;
        a.h_out += (size_t)f0 * DANBO_H_STRIDE;
        if (a.confd != nullptr) a.confd += (size_t)f0 * J;
    }
    const int ntiles = (n + A16_BM - 1) / A16_BM;
    const long spp = (long)(a.R / a.G) * a.S;

    // Tiles are handed out by a ticket counter (one atomic per workgroup and tile, issued a tile ahead of its use): a tile costs
    // between 1 500 and 60 000 ticks depending on how many bones its rows lie in, and with ~10 tiles per workgroup no static
    // assignment balances that -- the slowest wavefront of a round-robin schedule took 1.5 x (cull order) to 2.5 x (grouped rows,
    // whose windows alias with the stride) the mean (tools/micro_assign.py --model).  Every workgroup starts on tile blockIdx.x;
    // tickets hand out the tiles from gridDim.x on.  A launch with no more tiles than workgroups (the training step's) draws
    // nothing.  The counter resets itself: every workgroup draws exactly (tiles it took by ticket) + A16_TICKETS_AHEAD tickets, so
    // atomicInc wraps to 0 on the last draw of the launch.  (Same-address atomics retire at ~17 ns each on this part: a draw per
    // tile is affordable, four per IDLE workgroup -- the first version -- cost the small training launches 35 us.)
    __shared__ unsigned s_ticket[2];
    const int nwg = (int)gridDim.x;
    if ((int)blockIdx.x >= ntiles) return;
    const bool dynamic = ntiles > nwg;                       // (grid-uniform)
    const unsigned wrap = (unsigned)(ntiles - nwg) + (unsigned)A16_TICKETS_AHEAD * (unsigned)nwg - 1u;
    unsigned pending = 0;             // thread 0: the ticket drawn one tile ahead
    int q0 = (int)blockIdx.x, q1 = ntiles, q2 = ntiles;      // this tile, the next (its inputs are being loaded), the one after
    if (dynamic) {
        if (tid == 0) {
            s_ticket[0] = atomicInc(a.ticket, wrap);
            s_ticket[1] = atomicInc(a.ticket, wrap);
            pending = atomicInc(a.ticket, wrap);
        }
        __syncthreads();
        q1 = nwg + (int)s_ticket[0];
        q2 = nwg + (int)s_ticket[1];
        __syncthreads();
    }
    // Per-tile inputs are fetched one tile ahead: list entry -> (valid bits, ray, depth) is a chain of two dependent HBM round
    // trips.  The record keeps what was LOADED (the point o + d z is formed when the tile starts): anything computed from the
    // loads here would make the compiler wait for them here -- the round-3 kernel formed the point at once and stalled a full
    // round trip per tile on its own "prefetch" (s_memtime trace: 2 000 of a tile's 7 500 ticks).
    struct TileIn {
        int ms, first_ms;
        uint32_t bits;
        float o[3], d[3], z;
    };
    const bool multi_pose = a.G > 1;      // (one pose: every row's pose index is 0, no look-up of the tile's first row)
    const unsigned spp32 = (unsigned)spp;
    auto load_ms = [&](int tile_, int& first_) {
        const int t_ = tile_ < ntiles ? tile_ : ntiles - 1;
        const int row_ = t_ * A16_BM + wave * 32 + m;
        const int rowc_ = row_ < n ? row_ : n - 1;
        first_ = 0;
        if (multi_pose) first_ = a.list ? a.list[t_ * A16_BM] : t_ * A16_BM;          // (tile starts are < n)
        return a.list ? a.list[rowc_] : rowc_;
    };
    auto load_rest = [&](int ms_, int first_, TileIn& t) {
        t.ms = ms_;
        t.first_ms = first_;
        t.bits = a.valid_bits[ms_];
        if (a.pts != nullptr) {
            t.o[0] = a.pts[3 * (size_t)ms_]; t.o[1] = a.pts[3 * (size_t)ms_ + 1]; t.o[2] = a.pts[3 * (size_t)ms_ + 2];
            t.d[0] = t.d[1] = t.d[2] = 0.f;
            t.z = 0.f;
        } else {
            const int r = (int)((unsigned)ms_ / (unsigned)a.S);
            t.o[0] = a.rays_o[3 * r]; t.o[1] = a.rays_o[3 * r + 1]; t.o[2] = a.rays_o[3 * r + 2];
            t.d[0] = a.rays_d[3 * r]; t.d[1] = a.rays_d[3 * r + 1]; t.d[2] = a.rays_d[3 * r + 2];
            t.z = a.z[ms_];
        }
    };
    int g_lds = -1;
    TileIn cur = {};
    int first_next = 0, ms_next = 0;
    if (q0 < ntiles) {      // (a workgroup without a tile -- n == 0 included -- touches nothing: its four draws above were its part)
        const int ms0 = load_ms(q0, first_next);
        load_rest(ms0, first_next, cur);
        ms_next = load_ms(q1, first_next);
    }

    // the weight fragments of the bone this wavefront evaluated last: layer 0 (<= 5 neighbour terms x hi, lo), layer 1 (2 k-steps x hi, lo)

    int tr = 0, it = 0;
    auto stamp = [&](int tag) {    // workgroup 7, wavefront 1, its tiles 2 .. 5: (tag, s_memtime) pairs
        if (a.trace != nullptr && blockIdx.x == 7 && wave == 1 && it >= 2 && it < 6 && lane == 0 && tr < 250) {
            a.trace[tr++] = tag;
            a.trace[tr++] = (long long)__builtin_amdgcn_s_memtime();
        }
    };
    for (; q0 < ntiles; ++it) {
        const int tile = q0;
        // the ticket drawn during the previous tile becomes the tile after the next two; the following draw is issued now and
        // read a tile from now (its latency is covered by this tile's work).  One barrier per tile: the four wavefronts of a
        // workgroup share the tile (and the pose staged in LDS).
        int q3 = ntiles;
        if (dynamic) {
            if (tid == 0) {
                s_ticket[it & 1] = pending;
                pending = atomicInc(a.ticket, wrap);
            }
            __syncthreads();
            q3 = nwg + (int)s_ticket[it & 1];
        }
        stamp(0);
        const int row = tile * A16_BM + wave * 32 + m;
        const bool row_ok = row < n;
        const int ms = cur.ms;
        const uint32_t bits = cur.bits;
        const int g = multi_pose ? (int)min((unsigned)ms / spp32, (unsigned)a.G - 1u) : 0;
        float pnt[3];
        if (a.pts != nullptr) { pnt[0] = cur.o[0]; pnt[1] = cur.o[1]; pnt[2] = cur.o[2]; }
        else sample_point(cur.o, cur.d, cur.z, pnt);
        // the pose of the tile's first row: its volumes (23 KB) and transforms are staged in LDS -- the 30 interpolation
        // taps per bone then cost LDS latency instead of an L2 round trip
        const int g_tile = multi_pose ? (int)min((unsigned)__builtin_amdgcn_readfirstlane(cur.first_ms) / spp32, (unsigned)a.G - 1u) : 0;
        if (g_tile != g_lds) {
            __syncthreads();  // nobody still reads the previous pose
            const float4* srcv = reinterpret_cast<const float4*>(a.volumes + (size_t)g_tile * J * VOL);
            for (int i = tid; i < J * VOL / 4; i += 256) reinterpret_cast<float4*>(s_vol)[i] = srcv[i];
            for (int i = tid; i < J * 16; i += 256) s_skt[i] = a.skts[(size_t)g_tile * J * 16 + i];
            g_lds = g_tile;
            __syncthreads();
            if (tid < J * 8) {          // eight threads per bone: max |v| over its 240 entries
                float mx = 0.f;
                for (int i = tid & 7; i < VOL; i += 8) mx = fmaxf(mx, fabsf(s_vol[(tid >> 3) * VOL + i]));
                mx = fmaxf(mx, __shfl_xor(mx, 1, 64));
                mx = fmaxf(mx, __shfl_xor(mx, 2, 64));
                mx = fmaxf(mx, __shfl_xor(mx, 4, 64));
                // (a NaN entry would be dropped by fmaxf: such a bone is never skipped)
                float bad = 0.f;
                for (int i = tid & 7; i < VOL; i += 8) bad += s_vol[(tid >> 3) * VOL + i] != s_vol[(tid >> 3) * VOL + i] ? 1.f : 0.f;
                bad += __shfl_xor(bad, 1, 64); bad += __shfl_xor(bad, 2, 64); bad += __shfl_xor(bad, 4, 64);
                if ((tid & 7) == 0) s_vmax[tid >> 3] = bad > 0.f ? INFINITY : mx;
            }
            __syncthreads();
        }
        // stage 2 of the NEXT tile (its list entry was requested a tile ago) and stage 1 of the one after
        TileIn nxt;
        load_rest(ms_next, first_next, nxt);
        ms_next = load_ms(q2, first_next);
        // ---------------------------------------------------------------- which bones matter here
        // A bone's logit only enters the blend where that bone is valid (p_j = s(a_j) * valid_j), so for this wavefront only
        // bones valid for >= 1 of its 32 samples are evaluated (all 24 when the caller wants confd) -- exact, not approximate.
        uint32_t todo = wave_or(row_ok ? (bits & ((1u << J) - 1u)) : 0u);
        if (a.confd != nullptr) todo = (1u << J) - 1u;
        stamp(1);
        float hacc[8];
        float qsum = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) hacc[e] = 0.f;
        while (todo != 0u) {            // wave-uniform
            const int j = __builtin_ctz(todo);
            todo &= todo - 1u;
            const int nq = __builtin_amdgcn_readfirstlane(s_nbi[j * A16_NBI + 5]);
            stamp(100 + j);
            // ---- the bone's weight fragments (L2-resident stream, 1 KB per piece = 16 B per lane), requested where their latency has
            //      cover and held no longer than needed (round 5: ten layer-0 fragments resident per bone + all six neighbours' B
            //      fragments kept for one block of MFMAs were 88 of the kernel's 216 registers): layer 1's four here, used behind the
            //      whole feature phase; layer 0's four per neighbour PAIR behind that pair's window test, used behind its gather
            const char* src = a.packed + (size_t)__builtin_amdgcn_readfirstlane(s_nbi[j * A16_NBI + 6]) * 1024 + lane * 16;
            half8 w1[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) w1[i] = *reinterpret_cast<const half8*>(src + (2 * nq + i) * 1024);
            // ---- layer 0 with the adjacency folded in; accumulator starts at the shared bias, every pair adds its terms as soon as
            //      its features exist (the same order q = 0, 1, 2, ... as one block behind the loop: the same bits) ----
            f32x16 acc;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const float4 b = *reinterpret_cast<const float4*>(s_b0 + 8 * jj + 4 * hh);
                acc[4 * jj] = b.x; acc[4 * jj + 1] = b.y; acc[4 * jj + 2] = b.z; acc[4 * jj + 3] = b.w;
            }
            // ---- features of the bone and its tree neighbours ----
            // lane half h evaluates neighbour 2t + h completely (transform, window, 15-feature gather -- the same
            // gather_bone_features() as K1b), then the halves trade 8 values so that lane (m, h) ends up with features
            // 8h .. 8h+7 of BOTH bones: the B-fragment layout.
            float fself[8];         // this lane's 8 features of the bone ITSELF in fp32: the blend h = sum_j p_j f_j uses these, not the
                                    // 22-bit hi + lo reconstruction (2.4e-7 of f, which sin(32 h) turns into 8e-6 of the MLP's inputs)
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                if (2 * t >= nq) break;     // wave-uniform
                const int qm = 2 * t + hh < nq ? 2 * t + hh : nq - 1;       // (an odd count: the last half repeats a neighbour, unused)
                const int jb = s_nbi[j * A16_NBI + qm];
                float f[16];
                const bool lds_route = g == g_lds;      // (else: a row of another pose inside this tile -- multi-pose chunks only)
                float x[3] = {0.f, 0.f, 0.f}, win = 0.f;
                bool needed = true;
                if (lds_route) {
                    win = a16_bone_coords(s_skt + 16 * jb, s_align + 16 * jb, s_scale + 4 * jb, s_inv + 4 * jb, pnt, x);
                    // A neighbour's features enter layer 0 as fp16 hi + lo pairs: a feature below 2^-25 is (0, 0) there.  Every feature
                    // of a bone is at most (largest |entry| of its volume) x window, so where that product is below 2^-26 for EVERY
                    // row whose logit of bone j is used (rows valid in j; all rows when the caller wants confd), both neighbours of
                    // this pair contribute exact zeros to the matrix products: no gather, no MFMAs -- the accumulators (which start at
                    // the bias) come out bit for bit the same.  A sample inside the forearm is beyond that distance of the upper arm
                    // unless it is near the elbow.
                    const bool used = row_ok && (a.confd != nullptr || ((bits >> j) & 1u) != 0u);
                    needed = used && !(mul_rn(win, s_vmax[jb]) < 0x1p-26f);
                }
                if (t >= 1 && !a.no_skip && !__any(needed)) continue;         // wave-uniform
                half8 wq[4];        // w0 of neighbours 2t and 2t + 1 (hi | lo each)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (4 * t + i < 2 * nq) wq[i] = *reinterpret_cast<const half8*>(src + (4 * t + i) * 1024);
                if (lds_route) {
                    a16_bone_gather(s_vol + jb * VOL, x, win, f);
                } else {  // through L1 / L2
                    float sk[12];
                    const float* srcp = a.skts + ((size_t)g * J + jb) * 16;
#pragma unroll
                    for (int q = 0; q < 12; ++q) sk[q] = srcp[q];
                    a16_bone_features(sk, s_align + 16 * jb, a.volumes + ((size_t)g * J + jb) * VOL, s_scale + 4 * jb, s_inv + 4 * jb, pnt, f);
                }
                f[15] = 0.f;
                float keep[8], recv[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    keep[e] = hh ? f[8 + e] : f[e];                       // my half of my own bone
                    recv[e] = lane_xor32(hh ? f[e] : f[8 + e]);           // my half of the partner's bone
                }
                // neighbour 2t: half 0 keeps, half 1 receives; neighbour 2t+1: the other way round
                float even[8], odd[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    even[e] = hh ? recv[e] : keep[e];
                    odd[e] = hh ? keep[e] : recv[e];
                }
                if (t == 0) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) fself[e] = even[e];
                }
                half8 fh, fl;
                a16_split8(even, fh, fl);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wq[0], fh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wq[0], fl, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wq[1], fh, acc, 0, 0, 0);
                if (2 * t + 1 < nq) {                                   // wave-uniform
                    a16_split8(odd, fh, fl);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wq[2], fh, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wq[2], fl, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wq[3], fh, acc, 0, 0, 0);
                }
            }
            stamp(200 + j);
            // ---- relu -> layer-1 B fragments ----
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = fmaxf(acc[r], 0.f);
            half8 zh[2], zl[2];
            a16_split8(v, zh[0], zl[0]);
            a16_split8(v + 8, zh[1], zl[1]);
            // ---- layer 1 ----
            const float* b1 = s_b1 + j * 32 + 4 * hh;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const float4 b = *reinterpret_cast<const float4*>(b1 + 8 * jj);
                acc[4 * jj] = b.x; acc[4 * jj + 1] = b.y; acc[4 * jj + 2] = b.z; acc[4 * jj + 3] = b.w;
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1[2 * ks], zh[ks], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1[2 * ks], zl[ks], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1[2 * ks + 1], zh[ks], acc, 0, 0, 0);
            }
            // ---- layer 2 (32 -> 1): half dot per lane, halves combined by one shuffle ----
            const float* w2 = s_w2 + j * 32 + 4 * hh;
            float part = 0.f;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const float4 w = *reinterpret_cast<const float4*>(w2 + 8 * jj);
                part = fmaf(fmaxf(acc[4 * jj], 0.f), w.x, part);
                part = fmaf(fmaxf(acc[4 * jj + 1], 0.f), w.y, part);
                part = fmaf(fmaxf(acc[4 * jj + 2], 0.f), w.z, part);
                part = fmaf(fmaxf(acc[4 * jj + 3], 0.f), w.w, part);
            }
            const float logit = (part + lane_xor32(part)) + s_b2[j];
            if (a.confd != nullptr && row_ok && hh == 0) a.confd[(size_t)row * J + j] = logit;
            // ---- masked sigmoid + blend of this lane's 8 features (neighbour 0 is the bone itself) ----
            const float valid = ((bits >> j) & 1u) ? 1.0f : 0.0f;
            // sigmoid through v_exp_f32 / v_rcp_f32 (1 ulp each; p enters h = sum p_j f_j, compared at 5e-6)
            const float sg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(logit * -1.44269504088896340736f));
            const float pj = (sg * 1.002f - 0.001f) * valid;
#pragma unroll
            for (int e = 0; e < 8; ++e) hacc[e] = fmaf(pj, fself[e], hacc[e]);
            if (TRAIN) qsum += pj;
        }
        stamp(3);
        if (row_ok) {
            float4* dst = reinterpret_cast<float4*>(a.h_out + (size_t)row * DANBO_H_STRIDE + 8 * hh);
            dst[0] = make_float4(hacc[0], hacc[1], hacc[2], hacc[3]);
            dst[1] = make_float4(hacc[4], hacc[5], hacc[6], hh ? (TRAIN ? qsum : 0.f) : hacc[7]);
        }
        cur = nxt;
        q0 = q1;
        q1 = q2;
        q2 = q3;
    }
}

}  // namespace danbo

using namespace danbo;

static long long* g_a16_trace = nullptr;
static int a16_no_skip() {
    static const int v = dev_env("DANBO_A16_NOSKIP", 0);
    return v;
}
/* dev tool: a buffer of 256 int64 that receives (tag, s_memtime) stamps of one wavefront of k_assign16 (tags: 0 tile start, 1 inputs
 * and bone masks ready, 2 features done, 100 + j before / 200 + j after the wait for bone j's weights, 3 all bones done); NULL: off */
extern "C" int danbo_assign16_set_trace(void* buf) {
    g_a16_trace = (long long*)buf;
    return 0;
}

extern "C" int danbo_assign16_pack(const float* w0, const float* adjw, const float* w1, void* packed16, void* stream) {
    DANBO_CHECK_ARG(w0 && adjw && w1 && packed16);
    A16PackArgs a = {w0, adjw, w1};
    hipLaunchKernelGGL(k_assign16_pack, dim3(512), dim3(256), 0, (hipStream_t)stream, a, reinterpret_cast<_Float16*>(packed16));
    DANBO_LAUNCH_RET();
}

extern "C" int danbo_gather_assign_blend16_fwd(const float* rays_o, const float* rays_d, const float* z,
                                                const float* pts, int R, int S, int G, const float* skts,
                                                const float* align, const float* axis_scale, const float* volumes,
                                                const uint32_t* valid_bits, const int32_t* list, const int32_t* count,
                                                int n, const void* packed16, const float* b0, const float* b1,
                                                const float* w2, const float* b2, float* h, float* confd,
                                                uint32_t* ticket, void* stream) {
    DANBO_CHECK_ARG(n >= 0 && valid_bits && h && packed16 && ticket && R > 0 && S > 0 && G > 0 && R % G == 0);
    DANBO_CHECK_ARG((z == nullptr) != (pts == nullptr));
    if (n == 0) return 0;
    A16Args a = {rays_o, rays_d, z, pts, R, S, G, skts, align, axis_scale, volumes, valid_bits, list, count, n,
                 reinterpret_cast<const char*>(packed16), b0, b1, w2, b2, h, confd, nullptr, g_a16_trace, ticket, a16_no_skip()};
    DANBO_ENSURE_LDS(k_assign16<false>, A16_LDS_BYTES);
    const int ntiles = ceil_div(n, A16_BM);
    const int grid = ntiles < 2 * num_cu() ? ntiles : 2 * num_cu();      // two workgroups per CU (launch bounds)
    hipLaunchKernelGGL(k_assign16<false>, dim3(grid), dim3(256), A16_LDS_BYTES, (hipStream_t)stream, a);
    DANBO_LAUNCH_RET();
}

extern "C" int danbo_gather_assign_blend16_train(const float* rays_o, const float* rays_d, const float* z, int R, int S, int G,
                                                 const float* skts, const float* align, const float* axis_scale,
                                                 const float* volumes, const uint32_t* valid_bits, const int32_t* list,
                                                 const int32_t* count, const int32_t* first, int n, const void* packed16,
                                                 const float* b0, const float* b1, const float* w2, const float* b2, float* h,
                                                 uint32_t* ticket, void* stream) {
    DANBO_CHECK_ARG(n >= 0 && valid_bits && h && packed16 && list && count && z && ticket && R > 0 && S > 0 && G > 0 && R % G == 0);
    if (n == 0) return 0;
    A16Args a = {rays_o, rays_d, z, nullptr, R, S, G, skts, align, axis_scale, volumes, valid_bits, list, count, n,
                 reinterpret_cast<const char*>(packed16), b0, b1, w2, b2, h, nullptr, first, nullptr, ticket, a16_no_skip()};
    DANBO_ENSURE_LDS(k_assign16<true>, A16_LDS_BYTES);
    const int ntiles = ceil_div(n, A16_BM);
    const int grid = ntiles < 2 * num_cu() ? ntiles : 2 * num_cu();
    hipLaunchKernelGGL(k_assign16<true>, dim3(grid), dim3(256), A16_LDS_BYTES, (hipStream_t)stream, a);
    DANBO_LAUNCH_RET();
}
