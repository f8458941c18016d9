// Row-wise glue kernels of the training step (everything between the hand-written stages that is not a GEMM):
// per-ray view inputs, the loss gradients, un-merging of the composite gradients, per-bone pair lists, Adam.
// (The positional encoding, the colour head and their adjoints live in the fused trunk kernels, k_mlp16.hip / k_mlp16_bwd.hip;
// the frame-code gradients in k_train_head.hip.)
// Reference: core/networks/nerf.py:176-209,252-279 (inference / encode_views), core/cutoff_embedder.py:62-73 (Embedder),
// core/trainer.py:396-422,507-536 (losses), torch.optim.Adam as configured by core/raycasters.py:75.  gfx950 only.
//
// Row layout of a step (R rays, S coarse + Sf importance samples):
//   rows [0, R)                     one "empty-space" row per ray: blended feature h = 0; every sample of the ray that lies in
//                                   no bone volume shares its raw output (exact: see DESIGN.md "Exact sparsity")
//   rows [R, R + n_c)               coarse samples inside >= 1 volume, in the order K1a compacted them
//   rows [R + n_c, R + n_c + n_f)   importance samples inside >= 1 volume
// cnt[] (device int32): [0] running compaction counter (n_c after the coarse cull, n_c + n_f after the second),
//   [1] n_c, [2] R + n_c, [3] n_f, [4] R + n_c + n_f, [5] n_c + n_f, [6] (R + n_c) rounded down to 128, [7] n_f + (R + n_c) % 128
//   ([1] .. [7] are derived by block 0 of the trunk's forward kernel of each pass, k_mlp16.hip)
#include "common.hpp"
#include "composite_bwd.hpp"

namespace danbo {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void atomic_max_abs(float* slot, float v) {
    if (v > 0.f) atomicMax(reinterpret_cast<unsigned*>(slot), __builtin_bit_cast(unsigned, v));
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}
// Workgroup-wide max / sum of per-thread values, then ONE atomic per workgroup: thousands of wavefronts raising the same word
// serialise in the L2 (k_train_rgb_head_bwd spent 190 us on two such words before this).  256-thread workgroups.
__device__ __forceinline__ void block_atomic_max(float* slot, float v, float* s_red /*[4]*/) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) atomic_max_abs(slot, fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3])));
}
__device__ __forceinline__ void block_atomic_add(float* slot, float v, float* s_red /*[4]*/) {
    v = wave_total(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float t = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
        if (t != 0.f) atomicAdd(slot, t);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// per-ray view inputs  vin[r] = [ PE_L(dir) | frame code | 0 ]   (reference nerf.py:252-279, encoders.py:179-189,570-578)
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_train_view_inputs(const float* __restrict__ rays_d, const float* __restrict__ skts, int R, int G,
                                                           int ray_mode, int normalise, int L, const float* __restrict__ codes,
                                                           int n_codes, int Cf, const int64_t* __restrict__ cam_idx,
                                                           float* __restrict__ vin, int ldv) {
    const int nd = 3 * (1 + 2 * L);
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < (long)R * ldv; idx += (long)gridDim.x * blockDim.x) {
        const int r = (int)(idx / ldv), c = (int)(idx % ldv);
        float out = 0.f;
        if (c < nd) {
            float d[3] = {rays_d[3 * r], rays_d[3 * r + 1], rays_d[3 * r + 2]};
            if (ray_mode == 1) {   // root_local: skts[g, 0, :3, :3] d   (sequential-k sum like torch.matmul)
                const float* m = skts + (size_t)(r / (R / G)) * J * 16;
                float t[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) t[k] = add_rn(add_rn(mul_rn(m[4 * k], d[0]), mul_rn(m[4 * k + 1], d[1])), mul_rn(m[4 * k + 2], d[2]));
                d[0] = t[0]; d[1] = t[1]; d[2] = t[2];
            }
            if (normalise) {       // F.normalize(p = 2, eps = 1e-12)
                const float nrm = fmaxf(norm3_torch(d[0], d[1], d[2]), 1e-12f);
                d[0] = div_rn(d[0], nrm); d[1] = div_rn(d[1], nrm); d[2] = div_rn(d[2], nrm);
            }
            if (c < 3) out = d[c];
            else {
                const int b = (c - 3) / 3, k = (c - 3) % 3, l = b >> 1;
                float sn, cs;
                pe_sincos(mul_rn(d[k], (float)(1 << l)), &sn, &cs);
                out = (b & 1) ? cs : sn;
            }
        } else if (c < nd + Cf && codes != nullptr) {
            long ci = cam_idx ? cam_idx[r] : 0;
            ci = ci < 0 ? 0 : (ci >= n_codes ? n_codes - 1 : ci);
            out = codes[ci * Cf + (c - nd)];
        }
        vin[idx] = out;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// loss gradients with respect to the composited maps of both passes (reference trainer.py:396-422: L1 / MSE on
// rgb + (1 - acc) * bg, mean over R x 3 entries), and the loss values (atomically summed into loss[0] fine, loss[1] coarse)
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_train_loss_grad(const float* __restrict__ rgb, const float* __restrict__ acc,
                                                         const float* __restrict__ rgb0, const float* __restrict__ acc0,
                                                         const float* __restrict__ target, const float* __restrict__ bgs, int use_bg,
                                                         int R, int mse, float w_fine, float w_coarse, float* __restrict__ g_rgb,
                                                         float* __restrict__ g_acc, float* __restrict__ g_rgb0,
                                                         float* __restrict__ g_acc0, float* __restrict__ loss) {
    float l_f = 0.f, l_c = 0.f;
    const float inv = 1.0f / (3.0f * (float)R);
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < R; r += gridDim.x * blockDim.x) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const float* c = p ? rgb0 : rgb;
            const float a = p ? acc0[r] : acc[r];
            const float wgt = p ? w_coarse : w_fine;
            float ga = 0.f, ls = 0.f;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float bg = use_bg ? (bgs ? bgs[3 * r + k] : 1.0f) : 0.f;
                const float e = c[3 * r + k] + (use_bg ? (1.0f - a) * bg : 0.f) - target[3 * r + k];
                const float g = (mse ? 2.0f * e : (e > 0.f ? 1.0f : (e < 0.f ? -1.0f : 0.f))) * inv * wgt;
                ls += mse ? e * e : fabsf(e);
                (p ? g_rgb0 : g_rgb)[3 * r + k] = g;
                ga -= g * bg;
            }
            (p ? g_acc0 : g_acc)[r] = ga;
            if (p) l_c += ls * inv * wgt; else l_f += ls * inv * wgt;
        }
    }
    __shared__ float s_red[4];
    block_atomic_add(loss + 0, l_f, s_red);
    block_atomic_add(loss + 1, l_c, s_red);
}

// ------------------------------------------------------------------------------------------------------------------
// Un-merge the gradient of the final composite (sorted order) onto the coarse / importance samples, add the coarse
// composite's own gradient, sum the samples outside every volume into the ray's empty-space row, and derive the
// soft-softmax labels (T_i alpha > 0, reference trainer.py:507-536) in un-sorted order.  One wavefront per ray.
// loss[2] += sum of label^2 over samples outside every volume (their assignment probability mass is 0).
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_train_draw_unmerge(float4* __restrict__ d_raw_c /*in: coarse composite's, out: combined*/,
                                                            const float4* __restrict__ d_raw_sorted, const int32_t* __restrict__ order,
                                                            const uint32_t* __restrict__ bits_c, const uint32_t* __restrict__ bits_f,
                                                            const float* __restrict__ weights, const float* __restrict__ alpha, int R,
                                                            int S, int Sf, float4* __restrict__ d_raw_f, float4* __restrict__ d_raw_rows,
                                                            uint8_t* __restrict__ label_c, uint8_t* __restrict__ label_f,
                                                            float* __restrict__ loss, float* __restrict__ maxabs) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
    const int St = S + Sf;
    float lsum = 0.f, mx = 0.f;
    for (int r = wave; r < R; r += nwaves) {
        float ex = 0.f, ey = 0.f, ez = 0.f, ew = 0.f;
        for (int i = lane; i < St; i += 64) {
            const size_t ms = (size_t)r * St + i;
            const int src = min(max(order[ms], 0), St - 1);
            float4 d = d_raw_sorted[ms];
            const uint8_t lab = (weights[ms] * alpha[ms] > 0.f) ? 1 : 0;
            bool inside;
            if (src < S) {
                const size_t q = (size_t)r * S + src;
                const float4 c = d_raw_c[q];
                d.x += c.x; d.y += c.y; d.z += c.z; d.w += c.w;
                d_raw_c[q] = d;
                label_c[q] = lab;
                inside = bits_c[q] != 0u;
            } else {
                const size_t q = (size_t)r * Sf + (src - S);
                d_raw_f[q] = d;
                label_f[q] = lab;
                inside = bits_f[q] != 0u;
            }
            if (!inside) { ex += d.x; ey += d.y; ez += d.z; ew += d.w; lsum += (float)lab; }
            else mx = fmaxf(mx, fmaxf(fmaxf(fabsf(d.x), fabsf(d.y)), fmaxf(fabsf(d.z), fabsf(d.w))));
        }
        ex = wave_total(ex); ey = wave_total(ey); ez = wave_total(ez); ew = wave_total(ew);
        if (lane == 0) d_raw_rows[r] = make_float4(ex, ey, ez, ew);
        mx = fmaxf(mx, fmaxf(fmaxf(fabsf(ex), fabsf(ey)), fmaxf(fabsf(ez), fabsf(ew))));
    }
    __shared__ float s_red[4];
    block_atomic_add(loss + 2, lsum, s_red);
    block_atomic_max(maxabs, mx, s_red);
}

// ------------------------------------------------------------------------------------------------------------------
// The step's four launches between the final composite and the input-gradient chain as ONE (round 5): per ray and wavefront the
// loss gradients (k_train_loss_grad's arithmetic on lane 0), the adjoints of the coarse and of the merged composite
// (composite_bwd_ray: k_composite_bwd's), and k_train_draw_unmerge -- the two composites' d raw stay in LDS between them instead
// of making a round trip through memory.  Every output is the four kernels' bit for bit; the loss SUMS are formed per wavefront
// instead of per thread (atomics either way).
// ------------------------------------------------------------------------------------------------------------------
struct MidArgs {
    const float *rgb, *acc, *rgb0, *acc0, *target, *bgs;
    int use_bg, R, S, Sf, mse;
    float w_fine, w_coarse, B;
    float *g_rgb, *g_acc, *g_rgb0, *g_acc0;                    // [R,3], [R]: kept (the workspace view exposes them)
    const float4 *raw_c, *raw_empty, *raw_sorted;
    const uint32_t *bits_c, *bits_f;
    const float *z_c, *z_sorted, *rays_d, *noise_c, *noise_f;
    const int32_t* order;
    const float *weights, *alpha;
    float4 *d_raw_c, *d_raw_f, *d_raw_rows;
    uint8_t *label_c, *label_f;
    float *loss, *maxabs;
};
constexpr int MID_MAX = 256;     // samples per ray and composite
__host__ __device__ inline int mid_lds_per_wave(int S, int Sf) { return ((2 * S + Sf) * 16 + (S + Sf) + 15) & ~15; }

// 16 wavefronts per workgroup: every workgroup ends in four atomics on the same four words (same-address atomics retire one after
// the other, ~17 ns each: with one wavefront per ray in workgroups of four the 768 x 4 atomics of a 3 072-ray batch were a third
// of the kernel), and the per-wavefront LDS rows are sized by the launch (S, S + Sf), not by the 256-sample maximum.
constexpr int MID_WAVES = 16;
template <int NC>
__global__ __launch_bounds__(64 * MID_WAVES) void k_train_mid(MidArgs a) {
    extern __shared__ __attribute__((aligned(16))) char mid_smem[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
    const int R = a.R, S = a.S, Sf = a.Sf, St = a.S + a.Sf;
    // this wavefront's rows: d raw of the coarse / merged samples, volume flags of the coarse / importance samples
    char* mine = mid_smem + (size_t)wv * mid_lds_per_wave(S, Sf);
    float4* const s_c_w = reinterpret_cast<float4*>(mine);
    float4* const s_m_w = s_c_w + S;
    uint8_t* const s_in_c_w = reinterpret_cast<uint8_t*>(s_m_w + St);
    uint8_t* const s_in_f_w = s_in_c_w + S;
    const float inv = 1.0f / (3.0f * (float)R);
    float l_f = 0.f, l_c = 0.f, lsum = 0.f, mx = 0.f;
    for (int r = wave; r < R; r += nwaves) {
        // ---- everything the ray needs from memory is requested up front (ONE round trip per ray instead of one per stage: the ray's
        //      chain of four dependent stages IS the kernel's duration, 12 wavefronts per CU): the loss inputs (every lane the same
        //      addresses), the un-merge's order / weights / alpha, the importance samples' volume bits, and -- inside the forward
        //      sweeps -- the two composites' raw, depths and noise
        float in_c[2][3], in_a[2], tg[3], bg[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            in_c[0][k] = a.rgb[3 * r + k];
            in_c[1][k] = a.rgb0[3 * r + k];
            tg[k] = a.target[3 * r + k];
            bg[k] = a.use_bg ? (a.bgs ? a.bgs[3 * r + k] : 1.0f) : 0.f;
        }
        in_a[0] = a.acc[r];
        in_a[1] = a.acc0[r];
        int ord[NC];
        float wal[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int i = c * 64 + lane;
            ord[c] = 0;
            wal[c] = 0.f;
            if (i < St) {
                const size_t ms = (size_t)r * St + i;
                ord[c] = a.order[ms];
                wal[c] = a.weights[ms] * a.alpha[ms];
            }
            if (i < Sf) s_in_f_w[i] = a.bits_f[(size_t)r * Sf + i] != 0u;
        }
        CompositeState<NC> st;          // ONE state at a time: coarse forward + backward sweep, then the merged composite's
        composite_fwd_sweep(a.raw_c, a.z_c, a.rays_d, r, S, a.B, a.noise_c, a.raw_empty, a.bits_c, lane, st);
#pragma unroll
        for (int c = 0; c < NC; ++c)
            if (c * 64 + lane < S) s_in_c_w[c * 64 + lane] = st.inside[c];
        // ---- loss gradients of the ray (reference trainer.py:396-422; k_train_loss_grad's arithmetic, every lane the same values)
        float g[2][4];      // [pass][rgb, acc]
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const float wgt = p ? a.w_coarse : a.w_fine;
            float ga = 0.f, ls = 0.f;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float e = in_c[p][k] + (a.use_bg ? (1.0f - in_a[p]) * bg[k] : 0.f) - tg[k];
                const float gk = (a.mse ? 2.0f * e : (e > 0.f ? 1.0f : (e < 0.f ? -1.0f : 0.f))) * inv * wgt;
                ls += a.mse ? e * e : fabsf(e);
                g[p][k] = gk;
                ga -= gk * bg[k];
            }
            g[p][3] = ga;
            if (lane == 0) {
                float* gc = p ? a.g_rgb0 : a.g_rgb;
                gc[3 * r] = g[p][0]; gc[3 * r + 1] = g[p][1]; gc[3 * r + 2] = g[p][2];
                (p ? a.g_acc0 : a.g_acc)[r] = ga;
                if (p) l_c += ls * inv * wgt; else l_f += ls * inv * wgt;
            }
        }
        // ---- adjoints of the two composites: d raw of the coarse samples and of the merged (sorted) samples, into LDS
        composite_bwd_sweep(st, S, a.B, g[1][0], g[1][1], g[1][2], g[1][3], lane, [&](int s, const float4& o) { s_c_w[s] = o; });
        composite_fwd_sweep(a.raw_sorted, a.z_sorted, a.rays_d, r, St, a.B, a.noise_f, nullptr, nullptr, lane, st);
        composite_bwd_sweep(st, St, a.B, g[0][0], g[0][1], g[0][2], g[0][3], lane, [&](int s, const float4& o) { s_m_w[s] = o; });
        // the LDS rows are this wavefront's own: its DS operations execute in order, the fence keeps the compiler from reordering them
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // ---- un-merge (k_train_draw_unmerge)
        float ex = 0.f, ey = 0.f, ez = 0.f, ew = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int i = c * 64 + lane;
            if (i >= St) continue;
            const int src = min(max(ord[c], 0), St - 1);
            float4 d = s_m_w[i];
            const uint8_t lab = (wal[c] > 0.f) ? 1 : 0;
            bool inside;
            if (src < S) {
                const size_t q = (size_t)r * S + src;
                const float4 cc = s_c_w[src];
                d.x += cc.x; d.y += cc.y; d.z += cc.z; d.w += cc.w;
                a.d_raw_c[q] = d;
                a.label_c[q] = lab;
                inside = s_in_c_w[src] != 0;
            } else {
                const size_t q = (size_t)r * Sf + (src - S);
                a.d_raw_f[q] = d;
                a.label_f[q] = lab;
                inside = s_in_f_w[src - S] != 0;
            }
            if (!inside) { ex += d.x; ey += d.y; ez += d.z; ew += d.w; lsum += (float)lab; }
            else mx = fmaxf(mx, fmaxf(fmaxf(fabsf(d.x), fabsf(d.y)), fmaxf(fabsf(d.z), fabsf(d.w))));
        }
        ex = wave_total(ex); ey = wave_total(ey); ez = wave_total(ez); ew = wave_total(ew);
        if (lane == 0) a.d_raw_rows[r] = make_float4(ex, ey, ez, ew);
        mx = fmaxf(mx, fmaxf(fmaxf(fabsf(ex), fabsf(ey)), fmaxf(fabsf(ez), fabsf(ew))));
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    // workgroup totals, then one atomic per word and workgroup
    __shared__ float s_red[4][MID_WAVES];
    l_f = wave_total(l_f); l_c = wave_total(l_c); lsum = wave_total(lsum); mx = wave_max(mx);
    if (lane == 0) { s_red[0][wv] = l_f; s_red[1][wv] = l_c; s_red[2][wv] = lsum; s_red[3][wv] = mx; }
    __syncthreads();
    if (threadIdx.x < 4) {
        float t = s_red[threadIdx.x][0];
#pragma unroll
        for (int w = 1; w < MID_WAVES; ++w) t = threadIdx.x == 3 ? fmaxf(t, s_red[3][w]) : t + s_red[threadIdx.x][w];
        if (threadIdx.x == 3) atomic_max_abs(a.maxabs, t);
        else if (t != 0.f) atomicAdd(a.loss + threadIdx.x, t);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// per-bone lists of (row) pairs: bone j's list holds every in-volume row whose sample lies inside bone j's volume
// (wave-aggregated appends: one atomic per wavefront and bone)
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_train_bone_lists(const uint32_t* __restrict__ bits_c, const uint32_t* __restrict__ bits_f,
                                                          const int32_t* __restrict__ row_sample, const int32_t* __restrict__ cnt, int R,
                                                          int cap, int32_t* __restrict__ lists, int32_t* __restrict__ cntb) {
    // per 256 rows: wave-level ballots -> per-wave counts in LDS -> ONE global atomic per bone for the workgroup (the 24 list
    // counters are hot words: one atomic per wavefront and bone serialised 4 000 of them, 50 us)
    __shared__ int s_cnt[J][4];
    __shared__ int s_base[J];
    const int rows = cnt[5], first_f = cnt[2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (long k0 = (long)blockIdx.x * blockDim.x; k0 < rows; k0 += (long)gridDim.x * blockDim.x) {
        const long k = k0 + threadIdx.x;
        const int i = R + (int)k;
        uint32_t b = 0;
        if (k < rows) b = i < first_f ? bits_c[row_sample[i]] : bits_f[row_sample[i]];
        unsigned long long bal[J];
#pragma unroll
        for (int j = 0; j < J; ++j) {
            bal[j] = __ballot((b >> j) & 1u);
            if (lane == 0) s_cnt[j][wave] = (int)__popcll(bal[j]);
        }
        __syncthreads();
        if (threadIdx.x < J) {
            const int j = threadIdx.x;
            int run = 0;
#pragma unroll
            for (int w = 0; w < 4; ++w) { const int c = s_cnt[j][w]; s_cnt[j][w] = run; run += c; }
            s_base[j] = run > 0 ? atomicAdd(cntb + j, run) : 0;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < J; ++j)
            if ((b >> j) & 1u)
                lists[(size_t)j * cap + s_base[j] + s_cnt[j][wave] + (int)__popcll(bal[j] & ((1ull << lane) - 1ull))] = i;
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Adam on the flat parameter buffer, exactly the update torch.optim.Adam (amsgrad = False, weight_decay = 0) performs:
//   m = lerp(m, g, 1 - b1);  v = b2 v + (1 - b2) g g;  p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps)
// lr, bc1 = 1 - b1^t, bc2s = sqrt(1 - b2^t), gs = gradient scale: kernel arguments (by value -- nothing the host can overwrite
// between the enqueue and the execution of the launch)
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                              float* __restrict__ v, long n, float lr, float bc1, float bc2s, float gs, float b1,
                                              float b2, float eps) {
    const float step = lr / bc1;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float gi = g[i] * gs;
        const float mi = m[i] + (1.0f - b1) * (gi - m[i]);      // torch lerp_: start + weight * (end - start) for weight < 0.5
        const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] = p[i] - step * (mi / (sqrtf(vi) / bc2s + eps));
    }
}

}  // namespace danbo

using namespace danbo;

// grid of a kernel whose workgroups each end in atomics on the same few words: two workgroups per CU instead of eight
static inline int few_grid(long items, int block) {
    const int g = stream_grid(items, block);
    return g < 2 * num_cu() ? g : 2 * num_cu();
}

extern "C" int danbo_train_view_inputs(const float* rays_d, const float* skts, int R, int G, int ray_mode, int normalise, int L_view,
                                       const float* codes, int n_codes, int Cf, const int64_t* cam_idx, float* vin, int ldv,
                                       void* stream) {
    DANBO_CHECK_ARG(rays_d && vin && R > 0 && G > 0 && R % G == 0 && L_view >= 0 && ldv % 4 == 0);
    DANBO_CHECK_ARG(ldv >= 3 * (1 + 2 * L_view) + (codes ? Cf : 0) && (ray_mode == 0 || skts));
    hipLaunchKernelGGL(k_train_view_inputs, dim3(stream_grid((long)R * ldv, 256)), dim3(256), 0, (hipStream_t)stream, rays_d, skts, R, G,
                       ray_mode, normalise, L_view, codes, n_codes, Cf, cam_idx, vin, ldv);
    DANBO_LAUNCH_RET();
}

extern "C" int danbo_train_loss_grad(const float* rgb, const float* acc, const float* rgb0, const float* acc0, const float* target,
                                     const float* bgs, int use_bg, int R, int mse, float w_fine, float w_coarse, float* g_rgb,
                                     float* g_acc, float* g_rgb0, float* g_acc0, float* loss, void* stream) {
    DANBO_CHECK_ARG(rgb && acc && rgb0 && acc0 && target && g_rgb && g_acc && g_rgb0 && g_acc0 && loss && R > 0);
    hipLaunchKernelGGL(k_train_loss_grad, dim3(few_grid(R, 256)), dim3(256), 0, (hipStream_t)stream, rgb, acc, rgb0, acc0, target, bgs,
                       use_bg, R, mse, w_fine, w_coarse, g_rgb, g_acc, g_rgb0, g_acc0, loss);
    DANBO_LAUNCH_RET();
}

extern "C" int danbo_train_draw_unmerge(float* d_raw_c, const float* d_raw_sorted, const int32_t* order, const uint32_t* bits_c,
                                        const uint32_t* bits_f, const float* weights, const float* alpha, int R, int S, int Sf,
                                        float* d_raw_f, float* d_raw_rows, uint8_t* label_c, uint8_t* label_f, float* loss,
                                        float* maxabs, void* stream) {
    DANBO_CHECK_ARG(d_raw_c && d_raw_sorted && order && bits_c && bits_f && weights && alpha && d_raw_f && d_raw_rows && label_c && label_f);
    DANBO_CHECK_ARG(loss && maxabs && R > 0 && S > 0 && Sf > 0);
    hipLaunchKernelGGL(k_train_draw_unmerge, dim3(few_grid((long)R * 64, 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<float4*>(d_raw_c), reinterpret_cast<const float4*>(d_raw_sorted), order, bits_c, bits_f, weights,
                       alpha, R, S, Sf, reinterpret_cast<float4*>(d_raw_f), reinterpret_cast<float4*>(d_raw_rows), label_c, label_f, loss,
                       maxabs);
    DANBO_LAUNCH_RET();
}

extern "C" int danbo_train_mid(const float* rgb, const float* acc, const float* rgb0, const float* acc0, const float* target, const float* bgs,
                               int use_bg, int R, int S, int Sf, int mse, float w_fine, float w_coarse, float B, float* g_rgb, float* g_acc,
                               float* g_rgb0, float* g_acc0, const float* raw_c, const float* raw_empty, const float* raw_sorted,
                               const uint32_t* bits_c, const uint32_t* bits_f, const float* z_c, const float* z_sorted, const float* rays_d,
                               const float* noise_c, const float* noise_f, const int32_t* order, const float* weights, const float* alpha,
                               float* d_raw_c, float* d_raw_f, float* d_raw_rows, uint8_t* label_c, uint8_t* label_f, float* loss,
                               float* maxabs, void* stream) {
    DANBO_CHECK_ARG(rgb && acc && rgb0 && acc0 && target && g_rgb && g_acc && g_rgb0 && g_acc0 && loss && maxabs);
    DANBO_CHECK_ARG(raw_c && raw_empty && raw_sorted && bits_c && bits_f && z_c && z_sorted && rays_d && order && weights && alpha);
    DANBO_CHECK_ARG(d_raw_c && d_raw_f && d_raw_rows && label_c && label_f && R > 0 && S > 0 && Sf > 0 && S + Sf <= MID_MAX && B > 0.f);
    MidArgs a{rgb, acc, rgb0, acc0, target, bgs, use_bg, R, S, Sf, mse, w_fine, w_coarse, B, g_rgb, g_acc, g_rgb0, g_acc0,
              reinterpret_cast<const float4*>(raw_c), reinterpret_cast<const float4*>(raw_empty), reinterpret_cast<const float4*>(raw_sorted),
              bits_c, bits_f, z_c, z_sorted, rays_d, noise_c, noise_f, order, weights, alpha, reinterpret_cast<float4*>(d_raw_c),
              reinterpret_cast<float4*>(d_raw_f), reinterpret_cast<float4*>(d_raw_rows), label_c, label_f, loss, maxabs};
    const int lds = MID_WAVES * mid_lds_per_wave(S, Sf);        // <= 16 x 8.5 KB
    const dim3 grid(stream_grid((long)R * 64, 64 * MID_WAVES)), block(64 * MID_WAVES);
    // (the LDS attribute is set once per instantiation: its largest launch)
#define DANBO_MID_LAUNCH(NC)                                                                           \
    do {                                                                                               \
        DANBO_ENSURE_LDS(k_train_mid<NC>, MID_WAVES * mid_lds_per_wave(64 * (NC), 0));                 \
        hipLaunchKernelGGL(k_train_mid<NC>, grid, block, lds, (hipStream_t)stream, a);                 \
    } while (0)
    switch ((S + Sf + 63) / 64) {
        case 1: DANBO_MID_LAUNCH(1); break;
        case 2: DANBO_MID_LAUNCH(2); break;
        case 3: DANBO_MID_LAUNCH(3); break;
        default: DANBO_MID_LAUNCH(4); break;
    }
#undef DANBO_MID_LAUNCH
    DANBO_LAUNCH_RET();
}

extern "C" int danbo_train_bone_lists(const uint32_t* bits_c, const uint32_t* bits_f, const int32_t* row_sample, const int32_t* cnt,
                                      int R, int rows_cap, int32_t* lists, int32_t* cntb, void* stream) {
    DANBO_CHECK_ARG(bits_c && bits_f && row_sample && cnt && lists && cntb && R >= 0 && rows_cap > 0);
    hipLaunchKernelGGL(k_train_bone_lists, dim3(stream_grid(rows_cap, 256)), dim3(256), 0, (hipStream_t)stream, bits_c, bits_f, row_sample,
                       cnt, R, rows_cap, lists, cntb);
    DANBO_LAUNCH_RET();
}

extern "C" int danbo_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long n, float lr, float bias_corr1,
                               float sqrt_bias_corr2, float grad_scale, float beta1, float beta2, float eps, void* stream) {
    DANBO_CHECK_ARG(params && grads && exp_avg && exp_avg_sq && n > 0 && bias_corr1 > 0.f && sqrt_bias_corr2 > 0.f);
    hipLaunchKernelGGL(k_adam, dim3(stream_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, params, grads, exp_avg, exp_avg_sq, n, lr,
                       bias_corr1, sqrt_bias_corr2, grad_scale, beta1, beta2, eps);
    DANBO_LAUNCH_RET();
}
