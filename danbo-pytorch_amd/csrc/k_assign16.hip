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
// sample (h=0: features 0..7, h=1: 8..14) and keeps them as fp16 hi/lo fragments (192 VGPRs);
// weights (236 KB in fragment order) stream through a 3-slot LDS ring shared by the workgroup.
#include "common.hpp"

namespace danbo {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int A16_CHUNK = 32768;
constexpr int A16_SLOTS = 3;
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
};

constexpr int A16_TABLE_FLOATS = 32 + J * 32 + J * 32 + J + J * 16 + J * 4 + 8 /*pad*/ + J * VOL + J * 16;  // + one pose's volumes and transforms
constexpr int A16_LDS_BYTES = A16_SLOTS * A16_CHUNK + ((A16_TABLE_FLOATS * 4 + 15) & ~15);

struct APipe {
    const char* packed;
    char* ring;
    int issue_chunk, issue_slot, cons_slot, wave, lane;
};

__device__ __forceinline__ void apipe_issue(APipe& p) {
    const char* src = p.packed + (size_t)p.issue_chunk * A16_CHUNK + p.wave * 8192 + p.lane * 16;
    char* dst = p.ring + p.issue_slot * A16_CHUNK + p.wave * 8192;
#pragma unroll
    for (int q = 0; q < 8; ++q)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + q * 1024),
                                         (__attribute__((address_space(3))) void*)(dst + q * 1024), 16, 0, 0);
    p.issue_chunk = p.issue_chunk + 1 == A16_NCHUNK ? 0 : p.issue_chunk + 1;
    p.issue_slot = p.issue_slot + 1 == A16_SLOTS ? 0 : p.issue_slot + 1;
}

__device__ __forceinline__ const char* apipe_begin(APipe& p) {
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    apipe_issue(p);
    const char* base = p.ring + p.cons_slot * A16_CHUNK + p.lane * 16;
    p.cons_slot = p.cons_slot + 1 == A16_SLOTS ? 0 : p.cons_slot + 1;
    return base;
}

__device__ __forceinline__ void a16_split8(const float* v, half8& hi, half8& lo) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const _Float16 hh = (_Float16)v[e];
        hi[e] = hh;
        lo[e] = (_Float16)(v[e] - (float)hh);
    }
}

__device__ __forceinline__ half8 a16_frag(const char* base, int piece_in_chunk) {
    return *reinterpret_cast<const half8*>(base + piece_in_chunk * 1024);
}

// TRAIN: device-side first row, and the pad slot h[15] receives q = sum_j p_j valid_j (the row's assignment mass, which the
// soft-softmax loss compares with its label -- reference core/trainer.py:507-536)
template <bool TRAIN>
__global__ __launch_bounds__(256, 1) void k_assign16(A16Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* s_b0 = reinterpret_cast<float*>(smem + A16_SLOTS * A16_CHUNK);  // [32]
    float* s_b1 = s_b0 + 32;                                               // [24][32]
    float* s_w2 = s_b1 + J * 32;                                           // [24][32]
    float* s_b2 = s_w2 + J * 32;                                           // [24]
    float* s_align = s_b2 + J;                                             // [24][16]
    float* s_scale = s_align + J * 16;                                     // [24][4]
    float* s_vol = s_scale + J * 4 + 8;                                    // [24][240] volumes of the staged pose (16-B aligned)
    float* s_skt = s_vol + J * VOL;                                        // [24][16]  its bone transforms

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = lane & 31, hh = lane >> 5;
    if (tid < 32) s_b0[tid] = a.b0[tid];
    for (int i = tid; i < J * 32; i += 256) { s_b1[i] = a.b1[i]; s_w2[i] = a.w2[i]; }
    if (tid < J) s_b2[tid] = a.b2[tid];
    for (int i = tid; i < J * 16; i += 256) s_align[i] = a.align[i];
    for (int i = tid; i < J * 4; i += 256) s_scale[i] = (i & 3) < 3 ? fabsf(a.axis_scale[(i >> 2) * 3 + (i & 3)]) : 1.f;
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
    if ((int)blockIdx.x >= ntiles) return;

    APipe p;
    p.packed = a.packed; p.ring = smem; p.issue_chunk = 0; p.issue_slot = 0; p.cons_slot = 0; p.wave = wave; p.lane = lane;
    apipe_issue(p);
    apipe_issue(p);
    const long spp = (long)(a.R / a.G) * a.S;

    // Per-tile inputs are fetched one tile ahead: list entry -> (valid bits, sample point) is a chain of two dependent
    // HBM round trips (about 9 000 cycles with nothing to hide them behind at one wavefront per SIMD).
    struct TileIn {
        int ms;
        uint32_t bits;
        float pnt[3];
    };
    auto load_ms = [&](int tile_) {
        const int row_ = tile_ * A16_BM + wave * 32 + m;
        const int rowc_ = row_ < n ? row_ : n - 1;
        return a.list ? a.list[rowc_] : rowc_;
    };
    auto load_rest = [&](int ms_, TileIn& t) {
        t.ms = ms_;
        t.bits = a.valid_bits[ms_];
        if (a.pts != nullptr) {
            t.pnt[0] = a.pts[3 * (size_t)ms_]; t.pnt[1] = a.pts[3 * (size_t)ms_ + 1]; t.pnt[2] = a.pts[3 * (size_t)ms_ + 2];
        } else {
            const int r = ms_ / a.S;
            const float o[3] = {a.rays_o[3 * r], a.rays_o[3 * r + 1], a.rays_o[3 * r + 2]};
            const float d[3] = {a.rays_d[3 * r], a.rays_d[3 * r + 1], a.rays_d[3 * r + 2]};
            sample_point(o, d, a.z[ms_], t.pnt);
        }
    };
    int g_lds = -1;
    TileIn cur;
    load_rest(load_ms(blockIdx.x), cur);
    int ms_next = load_ms(min((int)(blockIdx.x + gridDim.x), ntiles - 1));

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int row = tile * A16_BM + wave * 32 + m;
        const bool row_ok = row < n;
        const int ms = cur.ms;
        const uint32_t bits = cur.bits;
        const int g = (int)min((long)ms / spp, (long)a.G - 1);
        const float pnt[3] = {cur.pnt[0], cur.pnt[1], cur.pnt[2]};
        // the pose of the tile's first row: its volumes (23 KB) and transforms are staged in LDS -- the 30 interpolation
        // taps per bone then cost LDS latency instead of an L2 round trip that nothing hides at one wavefront per SIMD
        const int first_row = tile * A16_BM;
        const int first_ms = a.list ? a.list[first_row < n ? first_row : n - 1] : first_row;   // workgroup-uniform
        const int g_tile = (int)min((long)first_ms / spp, (long)a.G - 1);
        if (g_tile != g_lds) {
            __syncthreads();  // nobody still reads the previous pose
            const float4* srcv = reinterpret_cast<const float4*>(a.volumes + (size_t)g_tile * J * VOL);
            for (int i = tid; i < J * VOL / 4; i += 256) reinterpret_cast<float4*>(s_vol)[i] = srcv[i];
            for (int i = tid; i < J * 16; i += 256) s_skt[i] = a.skts[(size_t)g_tile * J * 16 + i];
            g_lds = g_tile;
            __syncthreads();
        }
        // stage 2 of the NEXT tile (its list entry was requested a tile ago) and stage 1 of the one after
        TileIn nxt;
        load_rest(ms_next, nxt);
        ms_next = load_ms(min(tile + 2 * (int)gridDim.x, ntiles - 1));
        // ---------------------------------------------------------------- which bones matter here
        // A bone's logit only enters the blend where that bone is valid (p_j = s(a_j) * valid_j), so
        // for this wavefront only bones valid for >= 1 of its 32 samples are evaluated (all 24 when
        // the caller wants confd); their layer-0 inputs are the features of those bones and of
        // their tree neighbours.  Everything else is skipped wave-uniformly -- exact, not approximate.
        uint32_t need_gnn = 0;
        {
            const uint32_t mybits = row_ok ? bits : 0u;
#pragma unroll
            for (int j = 0; j < J; ++j) need_gnn |= (__ballot((mybits >> j) & 1u) != 0ull ? 1u : 0u) << j;
            if (a.confd != nullptr) need_gnn = (1u << J) - 1u;
        }
        uint32_t need_feat = 0;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            uint32_t nbmask = 0;
#pragma unroll
            for (int q = 0; q <= a16_deg(j); ++q) nbmask |= 1u << a16_nb(j, q);
            if ((need_gnn >> j) & 1u) need_feat |= nbmask;
        }
        // ---------------------------------------------------------------- per-bone features
        // lane half h evaluates bones 2i + h completely (transform, window, 15-feature gather --
        // the same gather_bone_features() as K1b), then the halves trade 8 values so that lane
        // (m, h) ends up with features 8h..8h+7 of BOTH bones: the B-fragment layout.
        half8 fh[J], fl[J];
#pragma unroll
        for (int i = 0; i < J / 2; ++i) {
            if (((need_feat >> (2 * i)) & 3u) == 0u) continue;  // wave-uniform
            const int jb = 2 * i + hh;
            float pt[3], f[16];
            if (g == g_lds) {
                bone_local(s_skt + 16 * jb, s_align + 16 * jb, pnt, pt);
                gather_bone_features(s_vol + jb * VOL, pt, s_scale + 4 * jb, f);
            } else {  // a row of another pose inside this tile (multi-pose chunks only): through L1 / L2
                float sk[12];
                const float* src = a.skts + ((size_t)g * J + jb) * 16;
#pragma unroll
                for (int q = 0; q < 12; ++q) sk[q] = src[q];
                bone_local(sk, s_align + 16 * jb, pnt, pt);
                gather_bone_features(a.volumes + ((size_t)g * J + jb) * VOL, pt, s_scale + 4 * jb, f);
            }
            f[15] = 0.f;
            float keep[8], recv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                keep[e] = hh ? f[8 + e] : f[e];                       // my half of my own bone
                recv[e] = lane_xor32(hh ? f[e] : f[8 + e]);           // my half of the partner's bone
            }
            // bone 2i: half 0 keeps, half 1 receives; bone 2i+1: the other way round
            float even[8], odd[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                even[e] = hh ? recv[e] : keep[e];
                odd[e] = hh ? keep[e] : recv[e];
            }
            a16_split8(even, fh[2 * i], fl[2 * i]);
            a16_split8(odd, fh[2 * i + 1], fl[2 * i + 1]);
            // two bone pairs at a time: without a fence hipcc hoists every bone's loads and spills
            if (i & 1) __builtin_amdgcn_sched_barrier(0);
        }
        // ---------------------------------------------------------------- assignment GNN + blend
        float hacc[8];
        float qsum = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) hacc[e] = 0.f;
        const char* base = nullptr;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int p0 = a16_piece0(j);
            const int nq = a16_deg(j) + 1;
            const int pend = p0 + 2 * nq + 4;
            if (((need_gnn >> j) & 1u) == 0u) {
                // skipped bone: still take part in the weight ring (every wavefront must hit the
                // same barriers); at most one chunk boundary falls inside a bone's <= 14 pieces
                if (((p0 + 31) & ~31) < pend) base = apipe_begin(p);
                continue;
            }
            // ---- layer 0 with the adjacency folded in; accumulator starts at the shared bias ----
            f32x16 acc;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const float4 b = *reinterpret_cast<const float4*>(s_b0 + 8 * jj + 4 * hh);
                acc[4 * jj] = b.x; acc[4 * jj + 1] = b.y; acc[4 * jj + 2] = b.z; acc[4 * jj + 3] = b.w;
            }
#pragma unroll
            for (int q = 0; q < nq; ++q) {
                const int piece = p0 + 2 * q;
                if ((piece & 31) == 0) base = apipe_begin(p);
                const half8 ah = a16_frag(base, piece & 31), al = a16_frag(base, (piece & 31) + 1);
                const int jp = a16_nb(j, q);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, fh[jp], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, fl[jp], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, fh[jp], acc, 0, 0, 0);
            }
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
                const int piece = p0 + 2 * nq + 2 * ks;
                if ((piece & 31) == 0) base = apipe_begin(p);
                const half8 ah = a16_frag(base, piece & 31), al = a16_frag(base, (piece & 31) + 1);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, zh[ks], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, zl[ks], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, zh[ks], acc, 0, 0, 0);
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
            // ---- masked sigmoid + blend of this lane's 8 features ----
            const float valid = ((bits >> j) & 1u) ? 1.0f : 0.0f;
            const float pj = (sigmoidf_(logit) * 1.002f - 0.001f) * valid;
#pragma unroll
            for (int e = 0; e < 8; ++e) hacc[e] = fmaf(pj, (float)fh[j][e] + (float)fl[j][e], hacc[e]);
            if (TRAIN) qsum += pj;
            __builtin_amdgcn_sched_barrier(0);
        }
        if (row_ok) {
            float4* dst = reinterpret_cast<float4*>(a.h_out + (size_t)row * DANBO_H_STRIDE + 8 * hh);
            dst[0] = make_float4(hacc[0], hacc[1], hacc[2], hacc[3]);
            dst[1] = make_float4(hacc[4], hacc[5], hacc[6], hh ? (TRAIN ? qsum : 0.f) : hacc[7]);
        }
        cur = nxt;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}

}  // namespace danbo

using namespace danbo;

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
                                                void* stream) {
    DANBO_CHECK_ARG(n >= 0 && valid_bits && h && packed16 && R > 0 && S > 0 && G > 0 && R % G == 0);
    DANBO_CHECK_ARG((z == nullptr) != (pts == nullptr));
    if (n == 0) return 0;
    A16Args a = {rays_o, rays_d, z, pts, R, S, G, skts, align, axis_scale, volumes, valid_bits, list, count, n,
                 reinterpret_cast<const char*>(packed16), b0, b1, w2, b2, h, confd, nullptr};
    DANBO_ENSURE_LDS(k_assign16<false>, A16_LDS_BYTES);
    const int ntiles = ceil_div(n, A16_BM);
    const int grid = ntiles < num_cu() ? ntiles : num_cu();
    hipLaunchKernelGGL(k_assign16<false>, dim3(grid), dim3(256), A16_LDS_BYTES, (hipStream_t)stream, a);
    DANBO_LAUNCH_RET();
}

extern "C" int danbo_gather_assign_blend16_train(const float* rays_o, const float* rays_d, const float* z, int R, int S, int G,
                                                 const float* skts, const float* align, const float* axis_scale,
                                                 const float* volumes, const uint32_t* valid_bits, const int32_t* list,
                                                 const int32_t* count, const int32_t* first, int n, const void* packed16,
                                                 const float* b0, const float* b1, const float* w2, const float* b2, float* h,
                                                 void* stream) {
    DANBO_CHECK_ARG(n >= 0 && valid_bits && h && packed16 && list && count && z && R > 0 && S > 0 && G > 0 && R % G == 0);
    if (n == 0) return 0;
    A16Args a = {rays_o, rays_d, z, nullptr, R, S, G, skts, align, axis_scale, volumes, valid_bits, list, count, n,
                 reinterpret_cast<const char*>(packed16), b0, b1, w2, b2, h, nullptr, first};
    DANBO_ENSURE_LDS(k_assign16<true>, A16_LDS_BYTES);
    const int ntiles = ceil_div(n, A16_BM);
    const int grid = ntiles < num_cu() ? ntiles : num_cu();
    hipLaunchKernelGGL(k_assign16<true>, dim3(grid), dim3(256), A16_LDS_BYTES, (hipStream_t)stream, a);
    DANBO_LAUNCH_RET();
}
