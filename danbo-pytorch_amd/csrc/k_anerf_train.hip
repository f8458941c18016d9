// A-NeRF (nerf_type = nerf) on this library's own kernels END TO END: the per-ray view constants of the render path and every
// kernel of the TRAINING step that is not a dense W-wide layer (those are k_linear16 both ways and k_dw16).  gfx950 only.
//
//   k_anerf_view_consts        C[j, ray, :] = views_linears.0[:, view columns of joint j] . PE(unit local ray direction)      (render + training)
//   k_anerf_view_consts_bwd    d views_linears.0[:, view columns] = sum_rays PE(dir)^T dC                                       (slices + fixed-order sum)
//   k_anerf_color<TRAIN>       k_anerf.hip's colour head with a per-RAY table row and hv = relu(pre_v) kept for the backward
//   k_anerf_color_bwd          d raw -> d featv, d alpha (scaled for the fp16-split GEMMs), dC, d pre_ray, partial d rgb_linear
//   k_anerf_ray_table / k_anerf_code_grads / k_anerf_code_rows / k_anerf_code_scatter     the frame-code part of the view layer
//   k_anerf_relu_mask          dz_l = dY_l . [y_l > 0], re-centred by a power of two, running max
//   k_anerf_unmerge            d raw of the merged composite back onto the two passes' samples
// Reference: NeRF.forward / inference / encode_views (core/networks/nerf.py:107-122,176-209,252-279), CutoffEmbedder._embed with
// dist_inputs (core/cutoff_embedder.py:151-214), Optcodes (core/networks/embedding.py:17-39), Trainer.train_batch /
// compute_loss (core/trainer.py:257-302,348-422); what torch's autograd derives from them.
//
// Every sum over rays / samples that ends in a parameter gradient is taken in a FIXED order (slices written to scratch and added
// by one thread per entry, leaders instead of atomics for the frame codes): two runs of a step give the same bits.
#include "common.hpp"

namespace danbo {

constexpr int AV_TR = 64;          // rays per workgroup tile of the view-constant kernels
constexpr int AV_NK_MAX = 51;      // 3 (1 + 2 L), L <= 8
constexpr int AV_VW_MAX = 256;

// E[kk], kk = 3 b + axis: b = 0 the unit bone-local ray direction u, b = 1 + 2 l: sin(2^l u), 2 + 2 l: cos(2^l u) -- the values
// k_anerf_view_pe (k_anerf.hip) writes, in the order of views_linears.0's view columns of ONE joint
__device__ __forceinline__ void av_ray_pe(const float* __restrict__ rays_d, const float* __restrict__ skts, int R, int G, int L, int ray, int j,
                                          float* __restrict__ e /* stride 1 */) {
    const int rays_per_pose = R / G;
    const float* M = skts + ((size_t)min(ray / rays_per_pose, G - 1) * J + j) * 16;
    const float d[3] = {rays_d[3 * ray], rays_d[3 * ray + 1], rays_d[3 * ray + 2]};
    float q[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) q[a] = add_rn(add_rn(mul_rn(M[4 * a], d[0]), mul_rn(M[4 * a + 1], d[1])), mul_rn(M[4 * a + 2], d[2]));
    const float den = fmaxf(norm3_torch(q[0], q[1], q[2]), 1e-12f);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float u = div_rn(q[k], den);
        e[k] = u;
        for (int l = 0; l < L; ++l) {
            float sn, cs;
            sincosf(mul_rn(u, (float)(1 << l)), &sn, &cs);
            e[(1 + 2 * l) * 3 + k] = sn;
            e[(2 + 2 * l) * 3 + k] = cs;
        }
    }
}

// grid (ray tiles, 24 joints), 256 threads: thread c owns column c of the joint's [nk, VW] weight slice (registers when NK is the
// compile-time 27 of the shipped multires_views = 4, LDS otherwise); the tile's 64 x nk encodings are LDS broadcasts
template <int NK>
__global__ __launch_bounds__(256) void k_anerf_view_consts(const float* __restrict__ rays_d, const float* __restrict__ skts, int R, int G,
                                                           int L, const float* __restrict__ wj /*[24][nk][VW]*/, int VW,
                                                           float* __restrict__ C /*[24][R][VW]*/) {
    extern __shared__ __attribute__((aligned(16))) float s_av[];
    const int nk = 3 * (1 + 2 * L), nkp = (nk + 3) & ~3;
    float* s_e = s_av;                        // [AV_TR][nkp]
    float* s_w = s_av + AV_TR * nkp;          // [nk][VW] (NK == 0 only)
    const int j = blockIdx.y, tid = threadIdx.x;
    const float* w = wj + (size_t)j * nk * VW;
    float wr[NK > 0 ? NK : 1];
    if (NK > 0) {
#pragma unroll
        for (int k = 0; k < NK; ++k) wr[k] = tid < VW ? w[k * VW + tid] : 0.f;
    } else {
        for (int i = tid; i < nk * VW; i += 256) s_w[i] = w[i];
    }
    const int ntiles = (R + AV_TR - 1) / AV_TR;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        __syncthreads();
        const int ray0 = tile * AV_TR;
        if (tid < AV_TR) {
            float* e = s_e + tid * nkp;
            if (ray0 + tid < R) av_ray_pe(rays_d, skts, R, G, L, ray0 + tid, j, e);
            else for (int k = 0; k < nk; ++k) e[k] = 0.f;
            for (int k = nk; k < nkp; ++k) e[k] = 0.f;
        }
        __syncthreads();
        if (tid < VW) {
            const int live = min(AV_TR, R - ray0);
            float* out = C + ((size_t)j * R + ray0) * VW + tid;
            for (int r = 0; r < live; ++r) {
                const float* e = s_e + r * nkp;
                float acc = 0.f;
                if (NK > 0) {
#pragma unroll
                    for (int k4 = 0; k4 < (NK + 3) / 4; ++k4) {
                        const float4 ev = *reinterpret_cast<const float4*>(e + 4 * k4);
                        acc = fmaf(ev.x, wr[4 * k4], acc);
                        if (4 * k4 + 1 < NK) acc = fmaf(ev.y, wr[4 * k4 + 1 < NK ? 4 * k4 + 1 : 0], acc);
                        if (4 * k4 + 2 < NK) acc = fmaf(ev.z, wr[4 * k4 + 2 < NK ? 4 * k4 + 2 : 0], acc);
                        if (4 * k4 + 3 < NK) acc = fmaf(ev.w, wr[4 * k4 + 3 < NK ? 4 * k4 + 3 : 0], acc);
                    }
                } else {
                    for (int k = 0; k < nk; ++k) acc = fmaf(e[k], s_w[k * VW + tid], acc);
                }
                out[(size_t)r * VW] = acc;
            }
        }
    }
}

// wj[j][3 b + a][c] = views_w[c * ld + col0 + 72 b + 3 j + a]: the view columns of views_linears.0 regrouped per joint
__global__ __launch_bounds__(256) void k_anerf_wj_pack(const float* __restrict__ views_w, int ld, int col0, int VW, int nk, float* __restrict__ wj) {
    const long total = (long)J * nk * VW;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % VW), kk = (int)((i / VW) % nk), j = (int)(i / ((long)VW * nk));
        wj[i] = views_w[(size_t)c * ld + col0 + 72 * (kk / 3) + 3 * j + kk % 3];
    }
}

// partial[slice][j][kk][c] = sum over the slice's rays of E[ray][kk] dC[j][ray][c]; grid (slices, 24)
template <int NK>
__global__ __launch_bounds__(256) void k_anerf_view_consts_bwd(const float* __restrict__ rays_d, const float* __restrict__ skts, int R, int G,
                                                               int L, const float* __restrict__ dC /*[24][R][VW]*/, int VW, int slices,
                                                               float* __restrict__ partial /*[slices][24][nk][VW]*/) {
    extern __shared__ __attribute__((aligned(16))) float s_av[];
    const int nk = 3 * (1 + 2 * L), nkp = (nk + 3) & ~3;
    float* s_e = s_av;                        // [AV_TR][nkp]
    float* s_acc = s_av + AV_TR * nkp;        // [nk][VW] (NK == 0 only)
    const int j = blockIdx.y, sl = blockIdx.x, tid = threadIdx.x;
    const int per = ((R + slices - 1) / slices + AV_TR - 1) / AV_TR * AV_TR;
    const int ray_lo = sl * per, ray_hi = min(R, ray_lo + per);
    float acc[NK > 0 ? NK : 1];
    if (NK > 0) {
#pragma unroll
        for (int k = 0; k < NK; ++k) acc[k] = 0.f;
    } else {
        for (int i = tid; i < nk * VW; i += 256) s_acc[i] = 0.f;
    }
    for (int ray0 = ray_lo; ray0 < ray_hi; ray0 += AV_TR) {
        __syncthreads();
        if (tid < AV_TR) {
            float* e = s_e + tid * nkp;
            if (ray0 + tid < ray_hi) av_ray_pe(rays_d, skts, R, G, L, ray0 + tid, j, e);
            else for (int k = 0; k < nkp; ++k) e[k] = 0.f;
            for (int k = nk; k < nkp; ++k) e[k] = 0.f;
        }
        __syncthreads();
        if (tid < VW) {
            const int live = min(AV_TR, ray_hi - ray0);
            const float* src = dC + ((size_t)j * R + ray0) * VW + tid;
            for (int r = 0; r < live; ++r) {
                const float g = src[(size_t)r * VW];
                const float* e = s_e + r * nkp;
                if (NK > 0) {
#pragma unroll
                    for (int k = 0; k < NK; ++k) acc[k] = fmaf(e[k], g, acc[k]);
                } else {
                    for (int k = 0; k < nk; ++k) s_acc[k * VW + tid] = fmaf(e[k], g, s_acc[k * VW + tid]);
                }
            }
        }
    }
    if (tid < VW) {
        float* out = partial + (((size_t)sl * J + j) * nk) * VW + tid;
        if (NK > 0) {
#pragma unroll
            for (int k = 0; k < NK; ++k) out[(size_t)k * VW] = acc[k];
        } else {
            for (int k = 0; k < nk; ++k) out[(size_t)k * VW] = s_acc[k * VW + tid];
        }
    }
}

// g_views_w[c * ld + col0 + 72 b + 3 j + a] = sum_slices partial[.][j][3 b + a][c]   (slices added in order)
__global__ __launch_bounds__(256) void k_anerf_view_consts_bwd_reduce(const float* __restrict__ partial, int slices, int nk, int VW,
                                                                      float* __restrict__ g_views_w, int ld, int col0) {
    const long total = (long)J * nk * VW;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % VW), kk = (int)((i / VW) % nk), j = (int)(i / ((long)VW * nk));
        float s = 0.f;
        for (int q = 0; q < slices; ++q) s += partial[(size_t)q * total + i];
        g_views_w[(size_t)c * ld + col0 + 72 * (kk / 3) + 3 * j + kk % 3] = s;
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// colour head, training form (k_anerf.hip k_anerf_color: one wavefront per ray, the ray's 24 x VW joint vectors in registers):
//   pre_v[c] = featv[row][c] + table_ray[ray][c] + sum_j w[row][j] C[j][ray][c];  hv = relu(pre_v);  raw = (rgb_w hv + rgb_b, alpha[row])
// rows of the pass: row = rl * S + s (rl = ray - ray0).  hv [rows, VW] is kept for the backward.
__global__ __launch_bounds__(256) void k_anerf_color_train(const float* __restrict__ featv, int ldf, const float* __restrict__ w,
                                                           const float* __restrict__ C, const float* __restrict__ table_ray, int R_total,
                                                           int nrays, int S, int VW, const float* __restrict__ rgb_w,
                                                           const float* __restrict__ rgb_b, const float* __restrict__ alpha, int lda,
                                                           float* __restrict__ hv, float* __restrict__ raw_out) {
    const int lane = threadIdx.x & 63;
    const int wave_global = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    float rw[3][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = lane + 64 * i;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) rw[ch][i] = c < VW ? rgb_w[ch * VW + c] : 0.f;
    }
    const float rb0 = rgb_b[0], rb1 = rgb_b[1], rb2 = rgb_b[2];
    for (int ray = wave_global; ray < nrays; ray += nwaves) {
        float cj[J][4], tb[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = lane + 64 * i;
            tb[i] = c < VW ? table_ray[(size_t)ray * VW + c] : 0.f;
#pragma unroll
            for (int j = 0; j < J; ++j) cj[j][i] = c < VW ? C[((size_t)j * R_total + ray) * VW + c] : 0.f;
        }
        constexpr int U = 4;
        for (int s0 = 0; s0 < S; s0 += U) {
            float x[U][4], wj[U][J], al[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const size_t row = (size_t)ray * S + min(s0 + u, S - 1);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int c = lane + 64 * i;
                    x[u][i] = c < VW ? featv[row * ldf + c] : 0.f;
                }
#pragma unroll
                for (int j = 0; j < J; ++j) wj[u][j] = w[row * J + j];
                al[u] = alpha[row * lda];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
#pragma unroll
                for (int i = 0; i < 4; ++i) x[u][i] = (lane + 64 * i) < VW ? x[u][i] + tb[i] : 0.f;
#pragma unroll
                for (int j = 0; j < J; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i) x[u][i] = fmaf(wj[u][j], cj[j][i], x[u][i]);
                float pr = 0.f, pg = 0.f, pb = 0.f;
                const size_t row = (size_t)ray * S + s0 + u;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float xr = fmaxf(x[u][i], 0.f);
                    if (s0 + u < S && lane + 64 * i < VW) hv[row * VW + lane + 64 * i] = xr;
                    pr = fmaf(xr, rw[0][i], pr);
                    pg = fmaf(xr, rw[1][i], pg);
                    pb = fmaf(xr, rw[2][i], pb);
                }
                pr = wave_total(pr); pg = wave_total(pg); pb = wave_total(pb);
                if (lane == 0 && s0 + u < S)
                    reinterpret_cast<float4*>(raw_out)[row] = make_float4(pr + rb0, pg + rb1, pb + rb2, al[u]);
            }
        }
    }
}

// power of two that brings `maxabs` to [2^6, 2^7) (1 when maxabs is 0 / not finite); exact both ways
__device__ __forceinline__ float pow2_to_64(float maxabs) {
    const unsigned E = (__builtin_bit_cast(unsigned, maxabs) >> 23) & 255u;
    if (E == 0u || E == 255u) return 1.0f;
    int se = 127 + (127 + 6 - (int)E);
    se = se < 1 ? 1 : (se > 253 ? 253 : se);
    return __builtin_bit_cast(float, (unsigned)se << 23);
}

// One wavefront per ray over the rows of one pass.  d raw [rows, 4] (true scale) ->
//   g[c]      = (sum_ch d rgb_ch rgb_w[ch][c]) [hv[c] > 0]                       = d pre_v
//   d_featv   [rows, VW]  = g * sigma,   d_alpha_out[row * ld_da + 0..3] = (d raw[row][3] * sigma, 0, 0, 0)   (sigma = pow2_to_64(*raw_max): sig_top;
//             16 bytes: the column block behind the W feature gradients of a [rows, W + 4] buffer, its padding columns zeroed)
//   dC[j][ray][c] (+)= sum_s w[row][j] g[c],   d_pre_ray[ray][c] (+)= sum_s g[c]               (accumulate: the second pass adds)
//   part[wave][3][VW + 1]: this wavefront's sums of d rgb_ch hv[c] (+ [VW]: d rgb_ch) over all its rays
__global__ __launch_bounds__(256) void k_anerf_color_bwd(const float* __restrict__ d_raw, const float* __restrict__ hv, const float* __restrict__ w,
                                                         int R_total, int nrays, int S, int VW, const float* __restrict__ rgb_w,
                                                         const float* __restrict__ raw_max, float* __restrict__ sig_top,
                                                         float* __restrict__ d_featv, float* __restrict__ d_alpha_out, int ld_da,
                                                         float* __restrict__ dC, float* __restrict__ d_pre_ray, int accumulate,
                                                         float* __restrict__ part) {
    const int lane = threadIdx.x & 63;
    const int wave_global = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    const float sigma = pow2_to_64(*raw_max);
    if (blockIdx.x == 0 && threadIdx.x == 0) *sig_top = sigma;
    float rw[3][4], prw[3][4], prb[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = lane + 64 * i;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) { rw[ch][i] = c < VW ? rgb_w[ch * VW + c] : 0.f; prw[ch][i] = 0.f; }
    }
    for (int ray = wave_global; ray < nrays; ray += nwaves) {
        float acc[J][4], ap[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ap[i] = 0.f;
#pragma unroll
            for (int j = 0; j < J; ++j) acc[j][i] = 0.f;
        }
        constexpr int U = 2;
        for (int s0 = 0; s0 < S; s0 += U) {
            float h[U][4], wj[U][J];
            float4 dr[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const size_t row = (size_t)ray * S + min(s0 + u, S - 1);
#pragma unroll
                for (int i = 0; i < 4; ++i) h[u][i] = (lane + 64 * i) < VW ? hv[row * VW + lane + 64 * i] : 0.f;
#pragma unroll
                for (int j = 0; j < J; ++j) wj[u][j] = w[row * J + j];
                dr[u] = reinterpret_cast<const float4*>(d_raw)[row];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (s0 + u >= S) break;
                const size_t row = (size_t)ray * S + s0 + u;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float g = fmaf(dr[u].z, rw[2][i], fmaf(dr[u].y, rw[1][i], dr[u].x * rw[0][i]));
                    g = h[u][i] > 0.f ? g : 0.f;
                    if (lane + 64 * i < VW) d_featv[row * VW + lane + 64 * i] = g * sigma;
                    ap[i] += g;
#pragma unroll
                    for (int j = 0; j < J; ++j) acc[j][i] = fmaf(wj[u][j], g, acc[j][i]);
                    prw[0][i] = fmaf(dr[u].x, h[u][i], prw[0][i]);
                    prw[1][i] = fmaf(dr[u].y, h[u][i], prw[1][i]);
                    prw[2][i] = fmaf(dr[u].z, h[u][i], prw[2][i]);
                }
                prb[0] += dr[u].x; prb[1] += dr[u].y; prb[2] += dr[u].z;
                if (lane == 0) *reinterpret_cast<float4*>(d_alpha_out + row * ld_da) = make_float4(dr[u].w * sigma, 0.f, 0.f, 0.f);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = lane + 64 * i;
            if (c < VW) {
                float* pr = d_pre_ray + (size_t)ray * VW + c;
                *pr = accumulate ? *pr + ap[i] : ap[i];
#pragma unroll
                for (int j = 0; j < J; ++j) {
                    float* pc = dC + ((size_t)j * R_total + ray) * VW + c;
                    *pc = accumulate ? *pc + acc[j][i] : acc[j][i];
                }
            }
        }
    }
    float* out = part + (size_t)wave_global * 3 * (VW + 1);
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (lane + 64 * i < VW) out[ch * (VW + 1) + lane + 64 * i] = prw[ch][i];
        if (lane == 0) out[ch * (VW + 1) + VW] = prb[ch];
    }
}

// g_rgb_w[ch][c] = sum_waves part (both passes' blocks, in order), g_rgb_b[ch] likewise
__global__ __launch_bounds__(256) void k_anerf_rgb_reduce(const float* __restrict__ part, int nparts, int VW, float* __restrict__ g_rgb_w,
                                                          float* __restrict__ g_rgb_b) {
    const int total = 3 * (VW + 1);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int p = 0; p < nparts; ++p) s += part[(size_t)p * total + i];
        const int ch = i / (VW + 1), c = i % (VW + 1);
        if (c < VW) g_rgb_w[ch * VW + c] = s;
        else g_rgb_b[ch] = s;
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// frame codes: table_ray[ray][c] = views_b[c] + sum_k views_w[c][code0 + k] codes[cam(ray)][k]   (Optcodes in training: the row of
// the ray's camera, core/networks/embedding.py:24-39); no frame codes (code_size = 0): the bias alone
__global__ __launch_bounds__(256) void k_anerf_ray_table(const float* __restrict__ views_w, int ld, int code0, int code_size,
                                                         const float* __restrict__ views_b, const float* __restrict__ codes, int n_codes,
                                                         const int64_t* __restrict__ cam_idx, int R, int VW, float* __restrict__ table_ray) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (long)R * VW; i += (long)gridDim.x * blockDim.x) {
        const int ray = (int)(i / VW), c = (int)(i % VW);
        float acc = views_b[c];
        if (code_size > 0) {
            long cam = cam_idx[ray];
            cam = cam < 0 ? 0 : (cam >= n_codes ? n_codes - 1 : cam);
            const float* cd = codes + (size_t)cam * code_size;
            const float* wr = views_w + (size_t)c * ld + code0;
            for (int k = 0; k < code_size; ++k) acc = fmaf(wr[k], cd[k], acc);
        }
        table_ray[i] = acc;
    }
}

// d views_w[c][code0 + k] = sum_rays d_pre_ray[ray][c] codes[cam(ray)][k]  (k < code_size);  k == code_size: d views_b[c] = sum_rays
// d_pre_ray[ray][c].  One thread per (c, k), rays in order.
__global__ __launch_bounds__(256) void k_anerf_code_grads(const float* __restrict__ d_pre_ray, const float* __restrict__ codes, int n_codes,
                                                          int code_size, const int64_t* __restrict__ cam_idx, int R, int VW,
                                                          float* __restrict__ g_views_w, int ld, int code0, float* __restrict__ g_views_b) {
    const int per = code_size + 1;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < VW * per; i += gridDim.x * blockDim.x) {
        const int c = i / per, k = i % per;
        float s = 0.f;
        if (k == code_size) {
            for (int r = 0; r < R; ++r) s += d_pre_ray[(size_t)r * VW + c];
            g_views_b[c] = s;
        } else {
            for (int r = 0; r < R; ++r) {
                long cam = cam_idx[r];
                cam = cam < 0 ? 0 : (cam >= n_codes ? n_codes - 1 : cam);
                s = fmaf(d_pre_ray[(size_t)r * VW + c], codes[(size_t)cam * code_size + k], s);
            }
            g_views_w[(size_t)c * ld + code0 + k] = s;
        }
    }
}

// v[ray][k] = sum_c d_pre_ray[ray][c] views_w[c][code0 + k]: the ray's contribution to its camera's code gradient
__global__ __launch_bounds__(256) void k_anerf_code_rows(const float* __restrict__ d_pre_ray, const float* __restrict__ views_w, int ld, int code0,
                                                         int code_size, int R, int VW, float* __restrict__ v) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (long)R * code_size; i += (long)gridDim.x * blockDim.x) {
        const int ray = (int)(i / code_size), k = (int)(i % code_size);
        float s = 0.f;
        for (int c = 0; c < VW; ++c) s = fmaf(d_pre_ray[(size_t)ray * VW + c], views_w[(size_t)c * ld + code0 + k], s);
        v[i] = s;
    }
}

// g_codes[cam][k] = sum over the rays of that camera, in ray order, of v[ray][k] -- without atomics: the FIRST ray of a camera (its
// "leader": no earlier ray has the same index) adds up all of them.  One workgroup per ray; g_codes is zero on entry.
__global__ __launch_bounds__(128) void k_anerf_code_scatter(const float* __restrict__ v, const int64_t* __restrict__ cam_idx, int n_codes,
                                                            int code_size, int R, float* __restrict__ g_codes) {
    const int ray = blockIdx.x;
    long cam = cam_idx[ray];
    cam = cam < 0 ? 0 : (cam >= n_codes ? n_codes - 1 : cam);
    __shared__ int s_first;
    if (threadIdx.x == 0) s_first = 1;
    __syncthreads();
    for (int r = threadIdx.x; r < ray; r += blockDim.x) {
        long o = cam_idx[r];
        o = o < 0 ? 0 : (o >= n_codes ? n_codes - 1 : o);
        if (o == cam) s_first = 0;
    }
    __syncthreads();
    if (!s_first) return;
    for (int k = threadIdx.x; k < code_size; k += blockDim.x) {
        float s = 0.f;
        for (int r = ray; r < R; ++r) {
            long o = cam_idx[r];
            o = o < 0 ? 0 : (o >= n_codes ? n_codes - 1 : o);
            if (o == cam) s += v[(size_t)r * code_size + k];
        }
        g_codes[(size_t)cam * code_size + k] = s;
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// dz = t . [y > 0] . rho,  rho = pow2_to_64(*prev_max) / (what the producer of t was centred on): the backward GEMMs split their
// operands into fp16 hi + lo halves, so every gradient tensor is kept near 2^6 by an exact power of two; the factor follows the
// PREVIOUS layer's recorded maximum (one layer late: a dense layer moves the magnitude by a factor of a few, fp16 has 2^15 above
// and 2^-14 (normal lo halves: 2^-3) below) so that one pass over the data does mask, scale and maximum.
//   sig[out] = sig[in] * rho: the cumulative scale of dz against the true gradient (k_anerf_unscale divides it out of dW)
__global__ __launch_bounds__(256) void k_anerf_relu_mask(const float4* __restrict__ t, const float4* __restrict__ y, long n4,
                                                         const float* __restrict__ prev_max, const float* __restrict__ sig_in,
                                                         float* __restrict__ sig_out, float* __restrict__ max_out, float4* __restrict__ dz) {
    // prev_max holds max |dz_prev| = (true) * sig_in: re-centre to 2^6
    const float rho = prev_max ? pow2_to_64(*prev_max) : 1.0f;
    if (blockIdx.x == 0 && threadIdx.x == 0) *sig_out = *sig_in * rho;
    float m = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const float4 a = t[i], b = y[i];
        float4 o;
        o.x = b.x > 0.f ? a.x * rho : 0.f;
        o.y = b.y > 0.f ? a.y * rho : 0.f;
        o.z = b.z > 0.f ? a.z * rho : 0.f;
        o.w = b.w > 0.f ? a.w * rho : 0.f;
        dz[i] = o;
        m = fmaxf(fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fmaxf(fabsf(o.z), fabsf(o.w))), m);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    __shared__ float s_m[4];
    if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3]));
        if (m == m && m < 3.0e38f) atomicMax(reinterpret_cast<unsigned*>(max_out), __builtin_bit_cast(unsigned, m));   // >= 0: uint order
        else atomicMax(reinterpret_cast<unsigned*>(max_out), 0x7f7fffffu);
    }
}

// d_all[row] = d raw of the merged composite routed back through the sort order (+ the coarse composite's own gradient); rows:
// coarse sample (r, s) -> r S + s, importance sample (r, s) -> R S + r Sf + s.  Running max |d raw| -> max_out.
__global__ __launch_bounds__(256) void k_anerf_unmerge(const float4* __restrict__ d_sorted, const float4* __restrict__ d_c0,
                                                       const int32_t* __restrict__ order, int R, int S, int Sf, float4* __restrict__ d_all,
                                                       float* __restrict__ max_out) {
    const int T = S + Sf;
    float m = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (long)R * T; i += (long)gridDim.x * blockDim.x) {
        const int r = (int)(i / T);
        int idx = order[i];
        idx = idx < 0 ? 0 : (idx >= T ? T - 1 : idx);
        float4 g = d_sorted[i];
        long dst;
        if (idx < S) {
            dst = (long)r * S + idx;
            const float4 c = d_c0[dst];
            g.x += c.x; g.y += c.y; g.z += c.z; g.w += c.w;
        } else {
            dst = (long)R * S + (long)r * Sf + (idx - S);
        }
        d_all[dst] = g;
        m = fmaxf(fmaxf(fmaxf(fabsf(g.x), fabsf(g.y)), fmaxf(fabsf(g.z), fabsf(g.w))), m);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    __shared__ float s_m[4];
    if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3]));
        if (m == m && m < 3.0e38f) atomicMax(reinterpret_cast<unsigned*>(max_out), __builtin_bit_cast(unsigned, m));
        else atomicMax(reinterpret_cast<unsigned*>(max_out), 0x7f7fffffu);
    }
}

// [feature_linear.weight (W rows) ; alpha_linear.weight (1 row)] -> wstack [W + 1, K];  [feature_linear.bias ; alpha_linear.bias] -> bstack
__global__ __launch_bounds__(256) void k_anerf_stack_head(const float* __restrict__ fw, const float* __restrict__ fb, const float* __restrict__ aw,
                                                          const float* __restrict__ ab, int W, int K, float* __restrict__ wstack,
                                                          float* __restrict__ bstack) {
    const long total = (long)(W + 1) * K;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x)
        wstack[i] = i < (long)W * K ? fw[i] : aw[i - (long)W * K];
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i <= W; i += (long)gridDim.x * blockDim.x) bstack[i] = i < W ? fb[i] : ab[0];
}

// gradient segments computed from operands scaled by sig[idx]: g[r * ld + c] /= sig[idx]   (exact: powers of two)
struct AnerfSeg { float* p; int rows, cols, ld, sig; };
constexpr int AN_MAX_SEGS = 32;
struct AnerfSegs { AnerfSeg s[AN_MAX_SEGS]; int n; };
__global__ __launch_bounds__(256) void k_anerf_unscale(AnerfSegs segs, const float* __restrict__ sig) {
    const AnerfSeg s = segs.s[blockIdx.y];
    const float inv = 1.0f / sig[s.sig];
    const long total = (long)s.rows * s.cols;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        float* q = s.p + (i / s.cols) * s.ld + i % s.cols;
        *q *= inv;
    }
}

}  // namespace danbo

using namespace danbo;

// ================================================================================================================================
// C ABI: the building blocks
// ================================================================================================================================
static int av_lds_bytes(int L, int VW, bool generic) {
    const int nk = 3 * (1 + 2 * L), nkp = (nk + 3) & ~3;
    return (AV_TR * nkp + (generic ? nk * VW : 0)) * (int)sizeof(float);
}

extern "C" int danbo_anerf_view_wj_pack(const float* views_w, int ld, int col0, int VW, int L, float* wj, void* stream) {
    DANBO_CHECK_ARG(views_w && wj && VW >= 1 && VW <= AV_VW_MAX && L >= 0 && L <= 8 && col0 >= 0 && ld >= col0 + 72 * (1 + 2 * L));
    const int nk = 3 * (1 + 2 * L);
    hipLaunchKernelGGL(k_anerf_wj_pack, dim3(stream_grid((long)J * nk * VW, 256)), dim3(256), 0, (hipStream_t)stream, views_w, ld, col0, VW, nk, wj);
    DANBO_LAUNCH_RET();
}

extern "C" int danbo_anerf_view_consts_fwd(const float* rays_d, const float* skts, int R, int G, int L, const float* wj, int VW, float* C,
                                           void* stream) {
    DANBO_CHECK_ARG(rays_d && skts && wj && C && R > 0 && G > 0 && R % G == 0 && L >= 0 && L <= 8 && VW >= 1 && VW <= AV_VW_MAX);
    const int ntiles = ceil_div(R, AV_TR);
    const int gx = ntiles < num_cu() ? ntiles : num_cu();       // x 24 joints: a few workgroups per CU
    if (L == 4) {
        hipLaunchKernelGGL(k_anerf_view_consts<27>, dim3(gx, J), dim3(256), av_lds_bytes(L, VW, false), (hipStream_t)stream, rays_d, skts, R, G, L,
                           wj, VW, C);
    } else {
        DANBO_ENSURE_LDS(k_anerf_view_consts<0>, av_lds_bytes(8, AV_VW_MAX, true));
        hipLaunchKernelGGL(k_anerf_view_consts<0>, dim3(gx, J), dim3(256), av_lds_bytes(L, VW, true), (hipStream_t)stream, rays_d, skts, R, G, L,
                           wj, VW, C);
    }
    DANBO_LAUNCH_RET();
}

static int av_bwd_slices(int R) {
    int s = ceil_div(R, 4 * AV_TR);
    return s < 1 ? 1 : (s > 64 ? 64 : s);
}
extern "C" long danbo_anerf_view_consts_bwd_scratch_floats(int R, int L, int VW) {
    if (R < 1 || L < 0 || L > 8 || VW < 1 || VW > AV_VW_MAX) return -1;
    return (long)av_bwd_slices(R) * J * 3 * (1 + 2 * L) * VW;
}
extern "C" int danbo_anerf_view_consts_bwd(const float* rays_d, const float* skts, int R, int G, int L, const float* dC, int VW,
                                           float* g_views_w, int ld, int col0, float* scratch, void* stream) {
    DANBO_CHECK_ARG(rays_d && skts && dC && g_views_w && scratch && R > 0 && G > 0 && R % G == 0 && L >= 0 && L <= 8);
    DANBO_CHECK_ARG(VW >= 1 && VW <= AV_VW_MAX && col0 >= 0 && ld >= col0 + 72 * (1 + 2 * L));
    const int slices = av_bwd_slices(R), nk = 3 * (1 + 2 * L);
    if (L == 4) {
        hipLaunchKernelGGL(k_anerf_view_consts_bwd<27>, dim3(slices, J), dim3(256), av_lds_bytes(L, VW, false), (hipStream_t)stream, rays_d, skts,
                           R, G, L, dC, VW, slices, scratch);
    } else {
        DANBO_ENSURE_LDS(k_anerf_view_consts_bwd<0>, av_lds_bytes(8, AV_VW_MAX, true));
        hipLaunchKernelGGL(k_anerf_view_consts_bwd<0>, dim3(slices, J), dim3(256), av_lds_bytes(L, VW, true), (hipStream_t)stream, rays_d, skts, R,
                           G, L, dC, VW, slices, scratch);
    }
    hipLaunchKernelGGL(k_anerf_view_consts_bwd_reduce, dim3(stream_grid((long)J * nk * VW, 256)), dim3(256), 0, (hipStream_t)stream, scratch,
                       slices, nk, VW, g_views_w, ld, col0);
    DANBO_LAUNCH_RET();
}

static int color_grid(int nrays) {
    const int blocks = ceil_div(nrays, 4);
    return blocks < num_cu() * 4 ? blocks : num_cu() * 4;
}

extern "C" int danbo_anerf_color_train_fwd(const float* featv, int ld_featv, const float* w, const float* C, const float* table_ray,
                                           int R_total, int nrays, int S, int VW, const float* rgb_w, const float* rgb_b, const float* alpha,
                                           int ld_alpha, float* hv, float* raw_out, void* stream) {
    DANBO_CHECK_ARG(featv && w && C && table_ray && rgb_w && rgb_b && alpha && hv && raw_out && ld_featv >= VW && ld_alpha >= 1);
    DANBO_CHECK_ARG(VW > 0 && VW <= AV_VW_MAX && S > 0 && nrays >= 0 && nrays <= R_total);
    if (nrays == 0) return 0;
    hipLaunchKernelGGL(k_anerf_color_train, dim3(color_grid(nrays)), dim3(256), 0, (hipStream_t)stream, featv, ld_featv, w, C, table_ray, R_total,
                       nrays, S, VW, rgb_w, rgb_b, alpha, ld_alpha, hv, raw_out);
    DANBO_LAUNCH_RET();
}

extern "C" long danbo_anerf_color_bwd_part_floats(int nrays, int VW) {
    if (nrays < 1 || VW < 1 || VW > AV_VW_MAX) return -1;
    return (long)color_grid(nrays) * 4 * 3 * (VW + 1);
}
extern "C" int danbo_anerf_color_bwd(const float* d_raw, const float* hv, const float* w, int R_total, int nrays, int S, int VW,
                                     const float* rgb_w, const float* raw_max, float* sig_top, float* d_featv, float* d_alpha_out,
                                     int ld_dalpha, float* dC, float* d_pre_ray, int accumulate, float* part, void* stream) {
    DANBO_CHECK_ARG(d_raw && hv && w && rgb_w && raw_max && sig_top && d_featv && d_alpha_out && dC && d_pre_ray && part);
    DANBO_CHECK_ARG(VW > 0 && VW <= AV_VW_MAX && S > 0 && nrays >= 1 && nrays <= R_total && (uintptr_t)d_raw % 16 == 0);
    DANBO_CHECK_ARG(ld_dalpha >= 4 && ld_dalpha % 4 == 0 && (uintptr_t)d_alpha_out % 16 == 0);
    hipLaunchKernelGGL(k_anerf_color_bwd, dim3(color_grid(nrays)), dim3(256), 0, (hipStream_t)stream, d_raw, hv, w, R_total, nrays, S, VW, rgb_w,
                       raw_max, sig_top, d_featv, d_alpha_out, ld_dalpha, dC, d_pre_ray, accumulate, part);
    DANBO_LAUNCH_RET();
}
extern "C" int danbo_anerf_rgb_reduce(const float* part, long part_floats, int VW, float* g_rgb_w, float* g_rgb_b, void* stream) {
    DANBO_CHECK_ARG(part && g_rgb_w && g_rgb_b && VW > 0 && VW <= AV_VW_MAX && part_floats > 0 && part_floats % (3 * (VW + 1)) == 0);
    hipLaunchKernelGGL(k_anerf_rgb_reduce, dim3(ceil_div(3 * (VW + 1), 256)), dim3(256), 0, (hipStream_t)stream, part,
                       (int)(part_floats / (3 * (VW + 1))), VW, g_rgb_w, g_rgb_b);
    DANBO_LAUNCH_RET();
}

extern "C" int danbo_anerf_ray_table(const float* views_w, int ld, int code0, int code_size, const float* views_b, const float* codes,
                                     int n_codes, const int64_t* cam_idx, int R, int VW, float* table_ray, void* stream) {
    DANBO_CHECK_ARG(views_w && views_b && table_ray && R > 0 && VW > 0 && code_size >= 0 && ld >= code0 + code_size);
    DANBO_CHECK_ARG(code_size == 0 || (codes && cam_idx && n_codes > 0));
    hipLaunchKernelGGL(k_anerf_ray_table, dim3(stream_grid((long)R * VW, 256)), dim3(256), 0, (hipStream_t)stream, views_w, ld, code0, code_size,
                       views_b, codes, n_codes, cam_idx, R, VW, table_ray);
    DANBO_LAUNCH_RET();
}

/* d views_linears.0.weight[:, code columns], d views_linears.0.bias, d framecodes.codes.weight (zero on entry) from d_pre_ray [R, VW];
 * v: scratch [R, code_size] */
extern "C" int danbo_anerf_code_grads(const float* d_pre_ray, const float* views_w, int ld, int code0, int code_size, const float* codes,
                                      int n_codes, const int64_t* cam_idx, int R, int VW, float* g_views_w, float* g_views_b, float* g_codes,
                                      float* v, void* stream) {
    DANBO_CHECK_ARG(d_pre_ray && views_w && g_views_w && g_views_b && R > 0 && VW > 0 && code_size >= 0 && ld >= code0 + code_size);
    DANBO_CHECK_ARG(code_size == 0 || (codes && cam_idx && n_codes > 0 && g_codes && v));
    hipLaunchKernelGGL(k_anerf_code_grads, dim3(ceil_div((long)VW * (code_size + 1), 256)), dim3(256), 0, (hipStream_t)stream, d_pre_ray, codes,
                       n_codes, code_size, cam_idx, R, VW, g_views_w, ld, code0, g_views_b);
    if (code_size > 0) {
        hipLaunchKernelGGL(k_anerf_code_rows, dim3(stream_grid((long)R * code_size, 256)), dim3(256), 0, (hipStream_t)stream, d_pre_ray, views_w, ld,
                           code0, code_size, R, VW, v);
        hipLaunchKernelGGL(k_anerf_code_scatter, dim3(R), dim3(128), 0, (hipStream_t)stream, v, cam_idx, n_codes, code_size, R, g_codes);
    }
    DANBO_LAUNCH_RET();
}

extern "C" int danbo_anerf_relu_mask(const float* t, const float* y, long n, const float* prev_max, const float* sig_in, float* sig_out,
                                     float* max_out, float* dz, void* stream) {
    DANBO_CHECK_ARG(t && y && dz && sig_in && sig_out && max_out && n >= 0 && n % 4 == 0);
    DANBO_CHECK_ARG((uintptr_t)t % 16 == 0 && (uintptr_t)y % 16 == 0 && (uintptr_t)dz % 16 == 0);
    hipLaunchKernelGGL(k_anerf_relu_mask, dim3(stream_grid(n / 4 > 0 ? n / 4 : 1, 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4*>(t), reinterpret_cast<const float4*>(y), n / 4, prev_max, sig_in, sig_out, max_out,
                       reinterpret_cast<float4*>(dz));
    DANBO_LAUNCH_RET();
}

extern "C" int danbo_anerf_unmerge(const float* d_sorted, const float* d_c0, const int32_t* order, int R, int S, int Sf, float* d_all,
                                   float* max_out, void* stream) {
    DANBO_CHECK_ARG(d_sorted && d_c0 && order && d_all && max_out && R > 0 && S > 0 && Sf > 0);
    hipLaunchKernelGGL(k_anerf_unmerge, dim3(stream_grid((long)R * (S + Sf), 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4*>(d_sorted), reinterpret_cast<const float4*>(d_c0), order, R, S, Sf,
                       reinterpret_cast<float4*>(d_all), max_out);
    DANBO_LAUNCH_RET();
}

// ================================================================================================================================
// danbo_anerf_train_step: forward, losses and backward of one A-NeRF training batch behind ONE C call (host code: it enqueues
// kernels on `stream`; no allocation, no synchronisation -- capturable into a HIP graph).
//
// Rows: n = R (S + Sf); the coarse pass' samples are rows [0, R S), the importance pass' rows [R S, n): the two forward passes fill
// consecutive row ranges of the same buffers, the backward is ONE sweep over all rows.  Everything between the layers lives in HBM,
// row-major (k_linear16 / k_dw16 take rows of any 16-byte aligned stride):
//   x0 [n, in_ch]  density inputs (k_anerf_encode)            wcut [n, 24]  cutoff weights
//   y_l [n, W]     post-ReLU activations of pts_linears.l     head [n, W + 4] = [feature_linear(y_last) | alpha | pad]
//   featv [n, VW]  views_linears.0[:, :W] feature              hv [n, VW]    relu(view layer)
// Backward: d raw -> (k_anerf_color_bwd) d featv, d alpha, dC, d pre_ray -> d feature -> dY_last -> for l = D-1 .. 0: dz_l = dY_l
// [y_l > 0] (k_anerf_relu_mask), dY_{l-1} = dz_l W_l (k_linear16 on the transposed packing) -> all weight gradients in one k_dw16.
// Gradient operands are kept near 2^6 by exact powers of two (sig[]), divided out of the weight gradients at the end.
// ================================================================================================================================
namespace {

struct ACarver {
    char* base;
    size_t used;
    template <class T>
    T* take(size_t n) {
        used = (used + 255) & ~(size_t)255;
        T* p = base ? reinterpret_cast<T*>(base + used) : nullptr;
        used += n * sizeof(T);
        return p;
    }
};

constexpr int SIG_TOP = DANBO_ANERF_MAX_D;       // sig / mx slots: [0, D) the trunk layers, [SIG_TOP] d raw / the head
constexpr int AN_DW_SLICES = 16;

struct AShapes { int R, G, S, Sf, chunk; };

struct ABuffers {
    char* zero_begin;
    float *mx, *sig, *loss;
    char* zero_end;
    float *near, *far, *cyl_scratch, *z_c, *z_f, *z_sorted;
    int32_t* order;
    float *x0, *wcut, *y[DANBO_ANERF_MAX_D], *head, *featv, *hv, *raw_c, *raw_f, *raw_sorted, *weights0;
    float *wj, *C, *table_ray, *wstack, *bstack;
    char *pk_fwd[DANBO_ANERF_MAX_D], *pk_bwd[DANBO_ANERF_MAX_D], *pk_head, *pk_head_t, *pk_featv, *pk_featv_t;
    float *g_rgb, *g_acc, *g_rgb0, *g_acc0, *d_c0, *d_sorted, *d_all, *d_featv, *d_head, *t, *dz[DANBO_ANERF_MAX_D], *dC, *d_pre_ray, *part,
        *vc_scratch, *code_v, *dw_scratch;
    long part_floats_c, part_floats_f;
};

bool amodel_ok(const DanboAnerfTrainModel* m) {
    if (!m || m->D < 2 || m->D > DANBO_ANERF_MAX_D || m->W < 4 || m->W % 4 != 0 || m->W > 508 || m->VW < 4 || m->VW % 4 != 0 || m->VW > AV_VW_MAX)
        return false;
    if (m->skip < -1 || m->skip > m->D - 2 || m->L < 0 || m->L > 8 || m->L_view < 0 || m->L_view > 8 || m->code_size < 0) return false;
    for (int l = 0; l < m->D; ++l)
        if (!m->pts_w[l] || !m->pts_b[l] || !m->g_pts_w[l] || !m->g_pts_b[l]) return false;
    if (!m->alpha_w || !m->alpha_b || !m->feature_w || !m->feature_b || !m->views_w || !m->views_b || !m->rgb_w || !m->rgb_b) return false;
    if (!m->g_alpha_w || !m->g_alpha_b || !m->g_feature_w || !m->g_feature_b || !m->g_views_w || !m->g_views_b || !m->g_rgb_w || !m->g_rgb_b) return false;
    if (m->code_size > 0 && (!m->codes || !m->g_codes || m->n_codes < 1)) return false;
    if (!m->g_flat || m->n_flat < 1 || !m->align || !m->cutoff || !m->tau || !(m->density_scale > 0.f)) return false;
    return true;
}

inline int a_in_ch(const DanboAnerfTrainModel* m) { return J * (1 + 2 * m->L) + 3 * J; }
inline int a_ldv(const DanboAnerfTrainModel* m) { return m->W + 72 * (1 + 2 * m->L_view) + m->code_size; }
inline int a_k_of(const DanboAnerfTrainModel* m, int l) { return l == 0 ? a_in_ch(m) : (l == m->skip + 1 ? a_in_ch(m) + m->W : m->W); }

void a_describe_dw(const DanboAnerfTrainModel* m, const ABuffers& b, DanboDwLayer* L, int* n_layers) {
    const int W = m->W, D = m->D, ic = a_in_ch(m), ldh = W + 4;
    int k = 0;
    for (int l = 0; l < D; ++l) {
        DanboDwLayer d = DanboDwLayer{};
        d.dy = b.dz[l]; d.ldy = W; d.N = W; d.dy_maxabs = b.mx ? b.mx + l : nullptr;
        d.gw = m->g_pts_w[l]; d.gb = m->g_pts_b[l];
        if (l == 0) { d.x1 = b.x0; d.ld1 = ic; d.K1 = ic; }
        else if (l == m->skip + 1) { d.x1 = b.x0; d.ld1 = ic; d.K1 = ic; d.x2 = b.y[l - 1]; d.ld2 = W; d.K2 = W; }
        else { d.x1 = b.y[l - 1]; d.ld1 = W; d.K1 = W; }
        L[k++] = d;
    }
    DanboDwLayer& h = L[k++];      // [feature_linear ; alpha_linear] evaluated as one W + 1 wide layer
    h = DanboDwLayer{};
    h.dy = b.d_head; h.ldy = ldh; h.N = W + 1; h.x1 = b.y[D - 1]; h.ld1 = W; h.K1 = W;
    h.gw = m->g_feature_w; h.gb = m->g_feature_b; h.gw2 = m->g_alpha_w; h.gb2 = m->g_alpha_b; h.split_n = W;
    DanboDwLayer& f = L[k++];      // views_linears.0[:, :W] on feature_linear's output (its bias gradient comes with the frame codes')
    f = DanboDwLayer{};
    f.dy = b.d_featv; f.ldy = m->VW; f.N = m->VW; f.x1 = b.head; f.ld1 = ldh; f.K1 = W;
    f.gw = m->g_views_w; f.gw_ld = a_ldv(m); f.gw_col0 = 0;
    *n_layers = k;
}

ABuffers a_carve(ACarver& c, const AShapes& s, const DanboAnerfTrainModel* m, long dw_floats) {
    ABuffers b{};
    const int W = m->W, VW = m->VW, D = m->D, ic = a_in_ch(m), ldh = W + 4;
    const size_t R = (size_t)s.R, Mc = R * s.S, Mf = R * s.Sf, n = Mc + Mf;
    const int nk = 3 * (1 + 2 * m->L_view);
    c.used = (c.used + 255) & ~(size_t)255;
    b.zero_begin = c.base ? c.base + c.used : nullptr;
    b.mx = c.take<float>(16);
    b.sig = c.take<float>(16);
    b.loss = c.take<float>(8);
    b.zero_end = c.base ? c.base + c.used : (char*)c.used;
    b.near = c.take<float>(R);
    b.far = c.take<float>(R);
    b.cyl_scratch = c.take<float>(8 * (size_t)((s.R + s.chunk - 1) / s.chunk));
    b.z_c = c.take<float>(Mc);
    b.z_f = c.take<float>(Mf);
    b.z_sorted = c.take<float>(n);
    b.order = c.take<int32_t>(n);
    b.x0 = c.take<float>(n * ic);
    b.wcut = c.take<float>(n * J);
    for (int l = 0; l < D; ++l) b.y[l] = c.take<float>(n * W);
    b.head = c.take<float>(n * ldh);
    b.featv = c.take<float>(n * VW);
    b.hv = c.take<float>(n * VW);
    b.raw_c = c.take<float>(Mc * 4);
    b.raw_f = c.take<float>(Mf * 4);
    b.raw_sorted = c.take<float>(n * 4);
    b.weights0 = c.take<float>(Mc);
    b.wj = c.take<float>((size_t)J * nk * VW);
    b.C = c.take<float>((size_t)J * R * VW);
    b.table_ray = c.take<float>(R * VW);
    b.wstack = c.take<float>((size_t)(W + 1) * W);
    b.bstack = c.take<float>(W + 4);
    for (int l = 0; l < D; ++l) {
        const bool sk = l == m->skip + 1;
        b.pk_fwd[l] = c.take<char>((size_t)danbo_linear16_packed_bytes(W, sk || l == 0 ? ic : W, sk ? W : 0));
        b.pk_bwd[l] = l > 0 ? c.take<char>((size_t)danbo_linear16_packed_bytes(W, W, 0)) : nullptr;
    }
    b.pk_head = c.take<char>((size_t)danbo_linear16_packed_bytes(W + 1, W, 0));
    b.pk_head_t = c.take<char>((size_t)danbo_linear16_packed_bytes(W, W + 1, 0));
    b.pk_featv = c.take<char>((size_t)danbo_linear16_packed_bytes(VW, W, 0));
    b.pk_featv_t = c.take<char>((size_t)danbo_linear16_packed_bytes(W, VW, 0));
    b.g_rgb = c.take<float>(R * 3);
    b.g_acc = c.take<float>(R);
    b.g_rgb0 = c.take<float>(R * 3);
    b.g_acc0 = c.take<float>(R);
    b.d_c0 = c.take<float>(Mc * 4);
    b.d_sorted = c.take<float>(n * 4);
    b.d_all = c.take<float>(n * 4);
    b.d_featv = c.take<float>(n * VW);
    b.d_head = c.take<float>(n * ldh);
    b.t = c.take<float>(n * W);
    for (int l = 0; l < D; ++l) b.dz[l] = c.take<float>(n * W);
    b.dC = c.take<float>((size_t)J * R * VW);
    b.d_pre_ray = c.take<float>(R * VW);
    b.part_floats_c = b.part_floats_f = danbo_anerf_color_bwd_part_floats(s.R, VW);
    b.part = c.take<float>((size_t)(b.part_floats_c + b.part_floats_f));
    b.vc_scratch = c.take<float>((size_t)danbo_anerf_view_consts_bwd_scratch_floats(s.R, m->L_view, VW));
    b.code_v = c.take<float>(R * (size_t)(m->code_size > 0 ? m->code_size : 1));
    b.dw_scratch = c.take<float>((size_t)dw_floats);
    return b;
}

long a_dw_floats(const DanboAnerfTrainModel* m, const AShapes& s) {
    ACarver c0{nullptr, 0};
    ABuffers b0 = a_carve(c0, s, m, 0);
    DanboDwLayer L[DANBO_ANERF_MAX_D + 2];
    int nl = 0;
    a_describe_dw(m, b0, L, &nl);
    return danbo_dw16_scratch_floats(L, nl, AN_DW_SLICES);
}

bool a_shapes_ok(int R, int G, int S, int Sf, int chunk) {
    return R >= 1 && G >= 1 && R % G == 0 && S >= 3 && Sf >= 1 && S + Sf <= 256 && chunk >= 1 && (long)R * (S + Sf) < (1l << 30);
}

#define ANERF_TRY(call) do { const int rc_ = (call); if (rc_ != 0) return rc_; } while (0)

}  // namespace

extern "C" size_t danbo_anerf_train_workspace(const DanboAnerfTrainModel* m, int R, int G, int S, int Sf, int chunk) {
    if (!amodel_ok(m) || !a_shapes_ok(R, G, S, Sf, chunk)) return 0;
    AShapes s{R, G, S, Sf, chunk};
    ACarver c{nullptr, 0};
    a_carve(c, s, m, a_dw_floats(m, s));
    return c.used + 512;
}

extern "C" int danbo_anerf_train_workspace_view(const DanboAnerfTrainModel* m, int R, int G, int S, int Sf, int chunk, void* workspace,
                                                DanboTrainView* v) {
    DANBO_CHECK_ARG(amodel_ok(m) && workspace && v && a_shapes_ok(R, G, S, Sf, chunk));
    AShapes s{R, G, S, Sf, chunk};
    ACarver c{reinterpret_cast<char*>(((uintptr_t)workspace + 255) & ~(uintptr_t)255), 0};
    const ABuffers b = a_carve(c, s, m, 0);
    v->z_coarse = b.z_c; v->z_fine = b.z_f; v->z_sorted = b.z_sorted; v->order = b.order; v->bits_coarse = nullptr; v->bits_fine = nullptr;
    return 0;
}

extern "C" int danbo_anerf_train_step(const DanboAnerfTrainModel* m, const DanboTrainBatch* bt, const DanboTrainOut* o, void* workspace,
                                      size_t workspace_bytes, void* stream) {
    DANBO_CHECK_ARG(amodel_ok(m) && bt && o && workspace);
    const int R = bt->R, G = bt->G, S = bt->S, Sf = bt->Sf;
    DANBO_CHECK_ARG(a_shapes_ok(R, G, S, Sf, bt->chunk));
    DANBO_CHECK_ARG(bt->rays_o && bt->rays_d && bt->skts && bt->cyls && bt->target && (m->code_size == 0 || bt->cam_idx));
    DANBO_CHECK_ARG(bt->rng_state == nullptr || (bt->n_uniform >= 0 && bt->n_normal >= 0 && bt->n_uniform + bt->n_normal > 0 &&
                                                 (bt->n_uniform == 0 || bt->rng_uniform) && (bt->n_normal == 0 || bt->rng_normal)));
    DANBO_CHECK_ARG(o->rgb_map && o->disp_map && o->acc_map && o->alpha && o->weights && o->rgb0 && o->disp0 && o->acc0 && o->alpha0 && o->loss);
    DANBO_CHECK_ARG(workspace_bytes >= danbo_anerf_train_workspace(m, R, G, S, Sf, bt->chunk));
    hipStream_t st = (hipStream_t)stream;
    const AShapes sh{R, G, S, Sf, bt->chunk};
    ACarver c{reinterpret_cast<char*>(((uintptr_t)workspace + 255) & ~(uintptr_t)255), 0};
    const ABuffers b = a_carve(c, sh, m, a_dw_floats(m, sh));
    const int W = m->W, VW = m->VW, D = m->D, ic = a_in_ch(m), ldh = W + 4, ldv = a_ldv(m);
    const int view0 = W, code0 = W + 72 * (1 + 2 * m->L_view);
    const long Mc = (long)R * S, Mf = (long)R * Sf, n = Mc + Mf;
    const float B = m->density_scale;

    // ---- zero: running maxima, scales, loss terms; the flat gradient (frame-code rows of cameras that are not in the batch)
    zero_words(b.zero_begin, (long)((b.zero_end - b.zero_begin) / 4), m->g_flat, m->n_flat, st);
    if (bt->rng_state != nullptr)
        ANERF_TRY(danbo_random_draws(bt->rng_state, (long)bt->n_uniform, bt->rng_uniform, (long)bt->n_normal, bt->normal_std, bt->rng_normal, stream));

    // ---- packing (the weights changed in the last Adam step): every layer in the forward orientation, every layer whose input carries
    //      a gradient in the transposed one; [feature_linear ; alpha_linear] stacked; views_linears.0's view columns per joint
    for (int l = 0; l < D; ++l) {
        const int K = a_k_of(m, l);
        const bool sk = l == m->skip + 1;
        ANERF_TRY(danbo_linear16_pack(m->pts_w[l], K, 1, W, sk || l == 0 ? ic : W, sk ? W : 0, b.pk_fwd[l], stream));
        if (l > 0)      // dY_{l-1} = dz_l W_l[:, columns of y_{l-1}]:  (W_l^T)[n', k'] = W_l[k', col + n']
            ANERF_TRY(danbo_linear16_pack(m->pts_w[l] + (sk ? ic : 0), 1, K, W, W, 0, b.pk_bwd[l], stream));
    }
    hipLaunchKernelGGL(k_anerf_stack_head, dim3(stream_grid((long)(W + 1) * W, 256)), dim3(256), 0, st, m->feature_w, m->feature_b, m->alpha_w,
                       m->alpha_b, W, W, b.wstack, b.bstack);
    ANERF_TRY(danbo_linear16_pack(b.wstack, W, 1, W + 1, W, 0, b.pk_head, stream));
    ANERF_TRY(danbo_linear16_pack(b.wstack, 1, W, W, W + 1, 0, b.pk_head_t, stream));
    ANERF_TRY(danbo_linear16_pack(m->views_w, ldv, 1, VW, W, 0, b.pk_featv, stream));
    ANERF_TRY(danbo_linear16_pack(m->views_w, 1, ldv, W, VW, 0, b.pk_featv_t, stream));
    ANERF_TRY(danbo_anerf_view_wj_pack(m->views_w, ldv, view0, VW, m->L_view, b.wj, stream));

    // ---- per ray: bounds, stratified depths, view constants, bias + frame-code rows
    ANERF_TRY(danbo_near_far_cylinder(bt->rays_o, bt->rays_d, bt->cyls, R, G, 0.f, 1.f, bt->near_in, bt->far_in, bt->chunk, b.cyl_scratch, b.near,
                                      b.far, stream));
    ANERF_TRY(danbo_coarse_samples(b.near, b.far, R, S, bt->t_rand, b.z_c, stream));
    ANERF_TRY(danbo_anerf_view_consts_fwd(bt->rays_d, bt->skts, R, G, m->L_view, b.wj, VW, b.C, stream));
    ANERF_TRY(danbo_anerf_ray_table(m->views_w, ldv, code0, m->code_size, m->views_b, m->codes, m->n_codes, bt->cam_idx, R, VW, b.table_ray, stream));

    // ---- one network pass over the rows [r0, r0 + np) = the R x s samples at depths zz
    auto network = [&](const float* zz, int s, long r0, float* raw) -> int {
        const int np = (int)((long)R * s);
        float* x0 = b.x0 + r0 * ic;
        ANERF_TRY(danbo_anerf_encode_fwd_dtau(bt->rays_o, bt->rays_d, zz, nullptr, R, s, G, bt->skts, m->align, m->cutoff, m->tau, m->L, 0, np, x0,
                                              b.wcut + r0 * J, stream));
        for (int l = 0; l < D; ++l) {
            float* y = b.y[l] + r0 * W;
            if (l == 0) ANERF_TRY(danbo_linear16_fwd(x0, ic, ic, nullptr, 0, 0, b.pk_fwd[l], m->pts_b[l], W, 1, y, W, np, nullptr, stream));
            else if (l == m->skip + 1)
                ANERF_TRY(danbo_linear16_fwd(x0, ic, ic, b.y[l - 1] + r0 * W, W, W, b.pk_fwd[l], m->pts_b[l], W, 1, y, W, np, nullptr, stream));
            else ANERF_TRY(danbo_linear16_fwd(b.y[l - 1] + r0 * W, W, W, nullptr, 0, 0, b.pk_fwd[l], m->pts_b[l], W, 1, y, W, np, nullptr, stream));
        }
        float* head = b.head + r0 * ldh;
        ANERF_TRY(danbo_linear16_fwd(b.y[D - 1] + r0 * W, W, W, nullptr, 0, 0, b.pk_head, b.bstack, W + 1, 0, head, ldh, np, nullptr, stream));
        ANERF_TRY(danbo_linear16_fwd(head, ldh, W, nullptr, 0, 0, b.pk_featv, nullptr, VW, 0, b.featv + r0 * VW, VW, np, nullptr, stream));
        return danbo_anerf_color_train_fwd(b.featv + r0 * VW, VW, b.wcut + r0 * J, b.C, b.table_ray, R, R, s, VW, m->rgb_w, m->rgb_b, head + W, ldh,
                                           b.hv + r0 * VW, raw, stream);
    };
    ANERF_TRY(network(b.z_c, S, 0, b.raw_c));
    ANERF_TRY(danbo_composite_fwd(b.raw_c, b.z_c, bt->rays_d, R, S, B, bt->noise_c, o->rgb0, o->disp0, o->acc0, b.weights0, o->alpha0, stream));
    ANERF_TRY(danbo_importance_samples(b.z_c, b.weights0, R, S, Sf, bt->u_rand, b.z_f, b.z_sorted, b.order, stream));
    ANERF_TRY(network(b.z_f, Sf, Mc, b.raw_f));
    ANERF_TRY(danbo_merge_samples(b.raw_c, b.raw_f, b.order, R, S, Sf, 4, b.raw_sorted, stream));
    ANERF_TRY(danbo_composite_fwd(b.raw_sorted, b.z_sorted, bt->rays_d, R, S + Sf, B, bt->noise_f, o->rgb_map, o->disp_map, o->acc_map, o->weights,
                                  o->alpha, stream));

    // ---- losses (trainer.py:396-422), the adjoints of the two composites, the un-merge
    ANERF_TRY(danbo_train_loss_grad(o->rgb_map, o->acc_map, o->rgb0, o->acc0, bt->target, bt->bgs, m->use_background, R, m->loss_mse, m->rgb_loss_coef,
                                    m->rgb_loss_coef * m->coarse_weight, b.g_rgb, b.g_acc, b.g_rgb0, b.g_acc0, b.loss, stream));
    ANERF_TRY(danbo_composite_bwd(b.raw_c, b.z_c, bt->rays_d, R, S, B, bt->noise_c, b.g_rgb0, b.g_acc0, b.d_c0, stream));
    ANERF_TRY(danbo_composite_bwd(b.raw_sorted, b.z_sorted, bt->rays_d, R, S + Sf, B, bt->noise_f, b.g_rgb, b.g_acc, b.d_sorted, stream));
    ANERF_TRY(danbo_anerf_unmerge(b.d_sorted, b.d_c0, b.order, R, S, Sf, b.d_all, b.mx + SIG_TOP, stream));

    // ---- colour head and view branch
    ANERF_TRY(danbo_anerf_color_bwd(b.d_all, b.hv, b.wcut, R, R, S, VW, m->rgb_w, b.mx + SIG_TOP, b.sig + SIG_TOP, b.d_featv, b.d_head + W, ldh, b.dC,
                                    b.d_pre_ray, 0, b.part, stream));
    ANERF_TRY(danbo_anerf_color_bwd(b.d_all + Mc * 4, b.hv + Mc * VW, b.wcut + Mc * J, R, R, Sf, VW, m->rgb_w, b.mx + SIG_TOP, b.sig + SIG_TOP,
                                    b.d_featv + Mc * VW, b.d_head + Mc * ldh + W, ldh, b.dC, b.d_pre_ray, 1, b.part + b.part_floats_c, stream));
    ANERF_TRY(danbo_anerf_rgb_reduce(b.part, b.part_floats_c + b.part_floats_f, VW, m->g_rgb_w, m->g_rgb_b, stream));
    ANERF_TRY(danbo_anerf_view_consts_bwd(bt->rays_d, bt->skts, R, G, m->L_view, b.dC, VW, m->g_views_w, ldv, view0, b.vc_scratch, stream));
    ANERF_TRY(danbo_anerf_code_grads(b.d_pre_ray, m->views_w, ldv, code0, m->code_size, m->codes, m->n_codes, bt->cam_idx, R, VW, m->g_views_w,
                                     m->g_views_b, m->code_size > 0 ? m->g_codes : nullptr, b.code_v, stream));

    // ---- the input-gradient chain: d feature = d featv W_v[:, :W] -> dY_{D-1} = [d feature | d alpha] [W_f ; w_alpha] -> the trunk
    ANERF_TRY(danbo_linear16_fwd(b.d_featv, VW, VW, nullptr, 0, 0, b.pk_featv_t, nullptr, W, 0, b.d_head, ldh, (int)n, nullptr, stream));
    ANERF_TRY(danbo_linear16_fwd(b.d_head, ldh, W + 1, nullptr, 0, 0, b.pk_head_t, nullptr, W, 0, b.t, W, (int)n, nullptr, stream));
    for (int l = D - 1; l >= 0; --l) {
        const bool top = l == D - 1;
        ANERF_TRY(danbo_anerf_relu_mask(b.t, b.y[l], n * W, top ? nullptr : b.mx + l + 1, b.sig + (top ? SIG_TOP : l + 1), b.sig + l, b.mx + l, b.dz[l],
                                        stream));
        if (l > 0) ANERF_TRY(danbo_linear16_fwd(b.dz[l], W, W, nullptr, 0, 0, b.pk_bwd[l], nullptr, W, 0, b.t, W, (int)n, nullptr, stream));
    }

    // ---- all weight / bias gradients of the dense layers in one launch, then the powers of two divided out
    DanboDwLayer dwl[DANBO_ANERF_MAX_D + 2];
    int n_dw = 0;
    a_describe_dw(m, b, dwl, &n_dw);
    ANERF_TRY(danbo_dw16(dwl, n_dw, (int)n, nullptr, AN_DW_SLICES, b.dw_scratch, stream));
    AnerfSegs segs{};
    auto seg = [&](float* p, int rows, int cols, int ld, int sig) { segs.s[segs.n++] = AnerfSeg{p, rows, cols, ld, sig}; };
    for (int l = 0; l < D; ++l) {
        seg(m->g_pts_w[l], W, a_k_of(m, l), a_k_of(m, l), l);
        seg(m->g_pts_b[l], 1, W, W, l);
    }
    seg(m->g_feature_w, W, W, W, SIG_TOP); seg(m->g_feature_b, 1, W, W, SIG_TOP);
    seg(m->g_alpha_w, 1, W, W, SIG_TOP);   seg(m->g_alpha_b, 1, 1, 1, SIG_TOP);
    seg(m->g_views_w, VW, W, ldv, SIG_TOP);
    hipLaunchKernelGGL(k_anerf_unscale, dim3(64, segs.n), dim3(256), 0, st, segs, b.sig);
    // ---- loss terms for the caller: [0] rgb fine, [1] rgb coarse, [2] = [3] = 0
    hipLaunchKernelGGL(k_copy_words_, dim3(1), dim3(64), 0, st, reinterpret_cast<const uint32_t*>(b.loss), reinterpret_cast<uint32_t*>(o->loss), 4,
                       (const uint32_t*)nullptr, (uint32_t*)nullptr, 0);
    DANBO_LAUNCH_RET();
}
