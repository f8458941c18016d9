// K2: per-sample bone-assignment GNN (MixGNN 15->32 [tree-adjacency mix] ->32->1), masked
// sigmoid and feature blend.  gfx950 only.
//
// Mapping: one bone per lane.  A wavefront holds two samples (lanes 0..23 and 32..55); the
// per-bone weights of all three layers (24 x (15x32 + 32x32 + 32) fp32 = 147 KB) are resident
// in LDS for the whole persistent workgroup, laid out [k][c/4][bone][4] so that a wavefront's
// ds_read_b128 touches 24 consecutive 16-B slots (conflict-free).  The 70-nonzero skeleton
// adjacency is applied with wavefront shuffles between bone lanes; the blend
// h = sum_j p_j f_j is a 32-lane butterfly reduction.
#include "common.hpp"

namespace danbo {

constexpr int AW = 32;                          // agg_W
constexpr int ASSIGN_BLOCK = 512;
constexpr int ASSIGN_SPB = (ASSIGN_BLOCK / 64) * 2;  // samples per workgroup iteration
constexpr int W0P = FEAT * (AW / 4) * J * 4;    // 11520
constexpr int W1P = AW * (AW / 4) * J * 4;      // 24576
constexpr int W2P = (AW / 4) * J * 4;           // 768
constexpr int ASSIGN_LDS_FLOATS = W0P + W1P + W2P + W2P /*b1*/ + AW /*b0*/ + J /*b2*/ + J * 5 /*adj weights*/ +
                                  J * 16 /*align*/ + J * 4 /*|scale|*/;

// SMPL kinematic tree (core/utils/skeleton_utils.py:83-110): parent of each joint
__device__ __constant__ int8_t c_parent[J] = {0, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21};
// up to four graph neighbours (parent + children), -1 padded
__device__ __constant__ int8_t c_nbr[J][4] = {
    {1, 2, 3, -1},  {0, 4, -1, -1},  {0, 5, -1, -1},  {0, 6, -1, -1},  {1, 7, -1, -1},  {2, 8, -1, -1},
    {3, 9, -1, -1}, {4, 10, -1, -1}, {5, 11, -1, -1}, {6, 12, 13, 14}, {7, -1, -1, -1}, {8, -1, -1, -1},
    {9, 15, -1, -1}, {9, 16, -1, -1}, {9, 17, -1, -1}, {12, -1, -1, -1}, {13, 18, -1, -1}, {14, 19, -1, -1},
    {16, 20, -1, -1}, {17, 21, -1, -1}, {18, 22, -1, -1}, {19, 23, -1, -1}, {20, -1, -1, -1}, {21, -1, -1, -1}};

// geometry inputs of the fused (gather + assign) variant
struct GatherArgs {
    const float* rays_o;
    const float* rays_d;
    const float* z;
    const float* pts;
    int R, S, G;
    const float* skts;
    const float* align;
    const float* axis_scale;
    const float* volumes;
};

// FUSED = false: features are read from part_feat [n,24,15] (output of K1b).
// FUSED = true : each bone lane recomputes its own transform + factorised gather (K1b fused in;
//                the render path never materialises the 1440 B/sample part_feat tensor).
template <bool FUSED>
__global__ __launch_bounds__(ASSIGN_BLOCK) void k_assign_blend(
    GatherArgs ga,
    const float* __restrict__ part_feat, const uint32_t* __restrict__ valid_bits, const int32_t* __restrict__ list,
    const int32_t* __restrict__ count, int n_cap, const float* __restrict__ w0, const float* __restrict__ adjw,
    const float* __restrict__ b0, const float* __restrict__ w1, const float* __restrict__ b1,
    const float* __restrict__ w2, const float* __restrict__ b2, float* __restrict__ h_out,
    float* __restrict__ confd) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_w0 = smem;
    float* s_w1 = s_w0 + W0P;
    float* s_w2 = s_w1 + W1P;
    float* s_b1 = s_w2 + W2P;
    float* s_b0 = s_b1 + W2P;
    float* s_b2 = s_b0 + AW;
    float* s_adj = s_b2 + J;  // [24][5]: self, nbr0..3
    float* s_align = s_adj + J * 5;   // [24][16]   (FUSED only)
    float* s_scale = s_align + J * 16;  // [24][4]

    const int tid = threadIdx.x;
    // ---- stage + re-layout the weights ----
    for (int i = tid; i < J * FEAT * AW; i += ASSIGN_BLOCK) {  // w0 [j][k][c]
        const int c = i % AW, k = (i / AW) % FEAT, j = i / (AW * FEAT);
        s_w0[((k * (AW / 4) + (c >> 2)) * J + j) * 4 + (c & 3)] = w0[i];
    }
    for (int i = tid; i < J * AW * AW; i += ASSIGN_BLOCK) {  // w1 [j][c][c']
        const int c2 = i % AW, c = (i / AW) % AW, j = i / (AW * AW);
        s_w1[((c * (AW / 4) + (c2 >> 2)) * J + j) * 4 + (c2 & 3)] = w1[i];
    }
    for (int i = tid; i < J * AW; i += ASSIGN_BLOCK) {  // w2 [j][c], b1 [j][c]
        const int c = i % AW, j = i / AW;
        s_w2[((c >> 2) * J + j) * 4 + (c & 3)] = w2[i];
        s_b1[((c >> 2) * J + j) * 4 + (c & 3)] = b1[i];
    }
    if (FUSED) {
        for (int i = tid; i < J * 16; i += ASSIGN_BLOCK) s_align[i] = ga.align[i];
        for (int i = tid; i < J * 4; i += ASSIGN_BLOCK)
            s_scale[i] = (i & 3) < 3 ? fabsf(ga.axis_scale[(i >> 2) * 3 + (i & 3)]) : 1.f;
    }
    if (tid < AW) s_b0[tid] = b0[tid];
    if (tid < J) {
        s_b2[tid] = b2[tid];
        s_adj[tid * 5] = adjw[tid * J + tid];
        for (int q = 0; q < 4; ++q) {
            const int nb = c_nbr[tid][q];
            s_adj[tid * 5 + 1 + q] = nb >= 0 ? adjw[tid * J + nb] : 0.f;
        }
    }
    __syncthreads();

    const int n = resolve_count(count, n_cap);
    const int lane = tid & 63, wave = tid >> 6;
    const int slot = lane >> 5, jl = lane & 31;
    const bool bone_ok = jl < J;
    const int j = bone_ok ? jl : J - 1;
    int nb_lane[4];
    float nb_w[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int nb = c_nbr[j][q];
        nb_lane[q] = slot * 32 + (nb >= 0 ? nb : j);
        nb_w[q] = s_adj[j * 5 + 1 + q];
    }
    const float self_w = s_adj[j * 5];
    const float4* w0v = reinterpret_cast<const float4*>(s_w0) + j;
    const float4* w1v = reinterpret_cast<const float4*>(s_w1) + j;
    const float4* w2v = reinterpret_cast<const float4*>(s_w2) + j;
    const float4* b1v = reinterpret_cast<const float4*>(s_b1) + j;

    for (int row0 = blockIdx.x * ASSIGN_SPB; row0 < n; row0 += gridDim.x * ASSIGN_SPB) {
        const int row = row0 + wave * 2 + slot;
        const bool row_ok = row < n;
        const int rowc = row_ok ? row : n - 1;
        // ---- this bone's 15 gathered features ----
        float f[FEAT + 1];
        const int m = list ? list[rowc] : rowc;
        if (FUSED) {
            const long spp = (long)(ga.R / ga.G) * ga.S;
            const int g = (int)min((long)m / spp, (long)ga.G - 1);
            float p[3], pt[3], sk[12];
            if (ga.pts != nullptr) {
                p[0] = ga.pts[3 * (size_t)m]; p[1] = ga.pts[3 * (size_t)m + 1]; p[2] = ga.pts[3 * (size_t)m + 2];
            } else {
                const int r = m / ga.S;
                const float o[3] = {ga.rays_o[3 * r], ga.rays_o[3 * r + 1], ga.rays_o[3 * r + 2]};
                const float d[3] = {ga.rays_d[3 * r], ga.rays_d[3 * r + 1], ga.rays_d[3 * r + 2]};
                sample_point(o, d, ga.z[m], p);
            }
            const float* src = ga.skts + ((size_t)g * J + j) * 16;
#pragma unroll
            for (int i = 0; i < 12; ++i) sk[i] = src[i];
            bone_local(sk, s_align + 16 * j, p, pt);
            gather_bone_features(ga.volumes + ((size_t)g * J + j) * VOL, pt, s_scale + 4 * j, f);
            f[FEAT] = 0.f;
        } else {
            const float* src = part_feat + ((size_t)rowc * J + j) * FEAT;
#pragma unroll
            for (int k = 0; k < FEAT; ++k) f[k] = src[k];
            f[FEAT] = 0.f;
        }
        // ---- layer 0: per-bone 15 -> 32 ----
        // (rolled outer loops + register rotation keep every register index static while only
        //  one k-quad of weights is live; a fully unrolled body made hipcc hoist all 376 LDS
        //  reads and spill >1000 VGPRs)
        float y[AW];
#pragma unroll
        for (int c = 0; c < AW; ++c) y[c] = 0.f;
        {
            float fr[FEAT + 1];
#pragma unroll
            for (int k = 0; k < FEAT + 1; ++k) fr[k] = f[k];
#pragma unroll 1
            for (int kq = 0; kq < 4; ++kq) {
#pragma unroll
                for (int ki = 0; ki < 4; ++ki) {
                    const int kk = kq * 4 + ki;
                    const float fk = fr[ki];
#pragma unroll
                    for (int c4 = 0; c4 < AW / 4; ++c4) {
                        // k = 15 is a zero pad: clamp the address, the product is 0 * finite
                        const float4 w = w0v[((kk < FEAT ? kk : FEAT - 1) * (AW / 4) + c4) * J];
                        y[4 * c4 + 0] = fmaf(fk, w.x, y[4 * c4 + 0]);
                        y[4 * c4 + 1] = fmaf(fk, w.y, y[4 * c4 + 1]);
                        y[4 * c4 + 2] = fmaf(fk, w.z, y[4 * c4 + 2]);
                        y[4 * c4 + 3] = fmaf(fk, w.w, y[4 * c4 + 3]);
                    }
                }
#pragma unroll
                for (int k = 0; k < FEAT + 1 - 4; ++k) fr[k] = fr[k + 4];
            }
        }
        // ---- skeleton adjacency mix (self + parent/children) + shared bias + relu ----
#pragma unroll
        for (int c = 0; c < AW; ++c) {
            float acc = self_w * y[c];
#pragma unroll
            for (int q = 0; q < 4; ++q) acc = fmaf(nb_w[q], __shfl(y[c], nb_lane[q], 64), acc);
            y[c] = fmaxf(acc + s_b0[c], 0.f);
        }
        // ---- layer 1: per-bone 32 -> 32 + bias + relu ----
        float z1[AW];
#pragma unroll
        for (int c4 = 0; c4 < AW / 4; ++c4) {
            const float4 b = b1v[c4 * J];
            z1[4 * c4 + 0] = b.x; z1[4 * c4 + 1] = b.y; z1[4 * c4 + 2] = b.z; z1[4 * c4 + 3] = b.w;
        }
        {
            float yr[AW];
#pragma unroll
            for (int c = 0; c < AW; ++c) yr[c] = y[c];
#pragma unroll 1
            for (int cq = 0; cq < AW / 4; ++cq) {
#pragma unroll
                for (int ci = 0; ci < 4; ++ci) {
                    const float yc = yr[ci];
#pragma unroll
                    for (int c4 = 0; c4 < AW / 4; ++c4) {
                        const float4 w = w1v[((cq * 4 + ci) * (AW / 4) + c4) * J];
                        z1[4 * c4 + 0] = fmaf(yc, w.x, z1[4 * c4 + 0]);
                        z1[4 * c4 + 1] = fmaf(yc, w.y, z1[4 * c4 + 1]);
                        z1[4 * c4 + 2] = fmaf(yc, w.z, z1[4 * c4 + 2]);
                        z1[4 * c4 + 3] = fmaf(yc, w.w, z1[4 * c4 + 3]);
                    }
                }
#pragma unroll
                for (int c = 0; c < AW - 4; ++c) yr[c] = yr[c + 4];
            }
        }
        // ---- layer 2: 32 -> 1 ----
        float a = s_b2[j];
#pragma unroll
        for (int c4 = 0; c4 < AW / 4; ++c4) {
            const float4 w = w2v[c4 * J];
            a = fmaf(fmaxf(z1[4 * c4 + 0], 0.f), w.x, a);
            a = fmaf(fmaxf(z1[4 * c4 + 1], 0.f), w.y, a);
            a = fmaf(fmaxf(z1[4 * c4 + 2], 0.f), w.z, a);
            a = fmaf(fmaxf(z1[4 * c4 + 3], 0.f), w.w, a);
        }
        // ---- masked sigmoid + blend ----
        const uint32_t bits = valid_bits[m];
        const float valid = (bone_ok && ((bits >> j) & 1u)) ? 1.0f : 0.0f;
        const float p = (sigmoidf_(a) * 1.002f - 0.001f) * valid;
        float hsum[FEAT];
#pragma unroll
        for (int k = 0; k < FEAT; ++k) {
            float v = bone_ok ? p * f[k] : 0.f;
#pragma unroll
            for (int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
            hsum[k] = v;
        }
        if (row_ok) {
            if (confd != nullptr && bone_ok) confd[(size_t)row * J + j] = a;
            if (jl == 0) {
                float4* dst = reinterpret_cast<float4*>(h_out + (size_t)row * DANBO_H_STRIDE);
                dst[0] = make_float4(hsum[0], hsum[1], hsum[2], hsum[3]);
                dst[1] = make_float4(hsum[4], hsum[5], hsum[6], hsum[7]);
                dst[2] = make_float4(hsum[8], hsum[9], hsum[10], hsum[11]);
                dst[3] = make_float4(hsum[12], hsum[13], hsum[14], 0.f);
            }
        }
    }
}

}  // namespace danbo

using namespace danbo;

template <bool FUSED>
static int launch_assign(const GatherArgs& ga, const float* part_feat, const uint32_t* valid_bits, const int32_t* list,
                         const int32_t* count, int n, const float* w0, const float* adjw, const float* b0,
                         const float* w1, const float* b1, const float* w2, const float* b2, float* h, float* confd,
                         void* stream) {
    const size_t lds = sizeof(float) * ASSIGN_LDS_FLOATS;
    DANBO_ENSURE_LDS(k_assign_blend<FUSED>, lds);
    const int tiles = ceil_div(n, ASSIGN_SPB);
    const int grid = tiles < num_cu() ? tiles : num_cu();
    hipLaunchKernelGGL(k_assign_blend<FUSED>, dim3(grid), dim3(ASSIGN_BLOCK), lds, (hipStream_t)stream, ga, part_feat,
                       valid_bits, list, count, n, w0, adjw, b0, w1, b1, w2, b2, h, confd);
    DANBO_LAUNCH_RET();
}

extern "C" int danbo_assign_blend_fwd(const float* part_feat, const uint32_t* valid_bits, const int32_t* list,
                                       const int32_t* count, int n, const float* w0, const float* adjw, const float* b0,
                                       const float* w1, const float* b1, const float* w2, const float* b2, float* h,
                                       float* confd, void* stream) {
    DANBO_CHECK_ARG(n >= 0 && part_feat && valid_bits && h);
    if (n == 0) return 0;
    GatherArgs ga = {};
    return launch_assign<false>(ga, part_feat, valid_bits, list, count, n, w0, adjw, b0, w1, b1, w2, b2, h, confd, stream);
}

extern "C" int danbo_gather_assign_blend_fwd(const float* rays_o, const float* rays_d, const float* z, const float* pts,
                                              int R, int S, int G, const float* skts, const float* align,
                                              const float* axis_scale, const float* volumes,
                                              const uint32_t* valid_bits, const int32_t* list, const int32_t* count,
                                              int n, const float* w0, const float* adjw, const float* b0,
                                              const float* w1, const float* b1, const float* w2, const float* b2,
                                              float* h, float* confd, void* stream) {
    DANBO_CHECK_ARG(n >= 0 && valid_bits && h && R > 0 && S > 0 && G > 0 && R % G == 0);
    DANBO_CHECK_ARG((z == nullptr) != (pts == nullptr));
    if (n == 0) return 0;
    GatherArgs ga = {rays_o, rays_d, z, pts, R, S, G, skts, align, axis_scale, volumes};
    return launch_assign<true>(ga, nullptr, valid_bits, list, count, n, w0, adjw, b0, w1, b1, w2, b2, h, confd, stream);
}
