// A-NeRF (nerf_type = nerf): the per-sample encoders on either side of the W = 448
// trunk.  gfx950 only.  The trunk itself is eight plain dense layers and runs as library GEMMs on the
// rows these kernels produce (core/anerf_engine.py); what is specific to A-NeRF lives here:
//   k_anerf_encode    bone-local distance + unit direction per joint, joint-distance cutoff PE  -> [n,432]
//   k_anerf_view_pe   per-RAY positional encoding of the 24 bone-local unit ray directions      -> [R,648]
//   k_anerf_color     cutoff-weighted view term + frame code + ReLU + rgb head                  -> raw [n,4]
// The view layer's 648 per-sample inputs are PE(dir)[ray, joint] * w[sample, joint]: the product with
// views_linears.0 factorises into per-ray, per-joint vectors C[ray, j, :] (24 x 224 floats, made once
// per ray) and a 24-term weighted sum per sample -- 5 376 MACs instead of 145 152 per sample.
#include "common.hpp"

namespace danbo {

constexpr int AN_TS = 8;                      // samples per workgroup iteration
constexpr int AN_BLOCK = AN_TS * J;           // 192 threads: (sample, joint)
constexpr int AN_MAXL = 8;

// reference: SamplePointsEmbedder.encode_pts (encoders.py:424-450) -> RelDistEncoder / VecNormEncoder
// (:630-651, :774-795) -> CutoffEmbedder._embed (cutoff_embedder.py:151-214; cut_to_dist, cutoff_shift,
// cutoff_inputs) -> NeRF.encode_pts cat (nerf.py:222-250)
// COMPACT: instead of the 432-wide density input the kernel writes what danbo_linear16_fwd_enc recomputes it from, 576 B per row:
//   [24][4] floats (inp_j, sh_j, w_j, dir_j.x) = cutoff - distance, the shifted distance the sin / cos take, the cutoff weight, the
//   x component of the unit direction to joint j; then [24][2] floats (dir_j.y, dir_j.z)
constexpr int AN_ENC_FLOATS = 144;
constexpr int AN_SUB = 4;                     // COMPACT: sub-tiles per copy-out (32 rows x 576 B = 18 KB of LDS)
template <bool COMPACT>
__global__ __launch_bounds__(AN_BLOCK) void k_anerf_encode(const float* __restrict__ rays_o,
                                                           const float* __restrict__ rays_d,
                                                           const float* __restrict__ z, const float* __restrict__ pts,
                                                           int R, int S, int G, const float* __restrict__ skts,
                                                           const float* __restrict__ align,
                                                           const float* __restrict__ cutoff, float tau_arg, int L,
                                                           long row0, int nrows, float* __restrict__ x0,
                                                           float* __restrict__ wout, const float* __restrict__ tau_dev = nullptr) {
    // tau as a DEVICE scalar (the module's `tau` buffer) when the launch is part of a captured graph: update_tau changes it
    // every step (core/cutoff_embedder.py:221-223) and a kernel argument would be frozen at capture time
    const float tau = tau_dev ? *tau_dev : tau_arg;
    extern __shared__ __attribute__((aligned(16))) float s_row[];  // [AN_TS][in_ch]
    __shared__ float s_align[J * 16];
    const int in_ch = COMPACT ? AN_ENC_FLOATS : (1 + 2 * L) * J + 3 * J;
    const int tid = threadIdx.x;
    for (int i = tid; i < J * 16; i += AN_BLOCK) s_align[i] = align[i];
    const long spp = (long)(R / G) * S;
    const int sl = tid / J, j = tid % J;
    const float c = cutoff[j];
    const float two_over_c = div_rn(2.0f, c);
    // COMPACT: AN_SUB sub-tiles of AN_TS rows are staged before ONE copy-out (round 6: a 4.6 KB copy between two barriers per eight
    // rows left the kernel at 2.7 TB/s); the 432-wide rows of the other form fill the LDS tile with one sub-tile
    constexpr int SUB = COMPACT ? AN_SUB : 1;
    const int ntiles = (nrows + AN_TS * SUB - 1) / (AN_TS * SUB);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        __syncthreads();
#pragma unroll 1
        for (int sub = 0; sub < SUB; ++sub) {
        const int row = (tile * SUB + sub) * AN_TS + sl;
        if (row < nrows) {
            const long m = row0 + row;
            const int g = (int)min(m / spp, (long)G - 1);
            float p[3], pt[3], sk[12];
            if (pts) { p[0] = pts[3 * m]; p[1] = pts[3 * m + 1]; p[2] = pts[3 * m + 2]; }
            else {
                const long r = m / S;
                const float o[3] = {rays_o[3 * r], rays_o[3 * r + 1], rays_o[3 * r + 2]};
                const float d[3] = {rays_d[3 * r], rays_d[3 * r + 1], rays_d[3 * r + 2]};
                sample_point(o, d, z[m], p);
            }
            const float* src = skts + ((size_t)g * J + j) * 16;
#pragma unroll
            for (int i = 0; i < 12; ++i) sk[i] = src[i];
            bone_local(sk, s_align + 16 * j, p, pt);
            const float v = norm3_torch(pt[0], pt[1], pt[2]);
            const float den = fmaxf(v, 1e-12f);
            float* out = s_row + (sub * AN_TS + sl) * in_ch;
            const float w = sub_rn(1.0f, sigmoidf_(mul_rn(tau, sub_rn(v, c))));
            const float inp = sub_rn(c, v);
            const float sh = sub_rn(mul_rn(inp, two_over_c), 1.0f);
            if (COMPACT) {
                *reinterpret_cast<float4*>(out + 4 * j) = make_float4(inp, sh, w, div_rn(pt[0], den));
                *reinterpret_cast<float2*>(out + 4 * J + 2 * j) = make_float2(div_rn(pt[1], den), div_rn(pt[2], den));
            } else {
            out[j] = mul_rn(inp, w);
            for (int l = 0; l < L; ++l) {
                float sn, cs;
                sincosf(mul_rn(sh, (float)(1 << l)), &sn, &cs);
                out[(1 + 2 * l) * J + j] = mul_rn(sn, w);
                out[(2 + 2 * l) * J + j] = mul_rn(cs, w);
            }
            float* dir = out + (1 + 2 * L) * J + 3 * j;
            dir[0] = div_rn(pt[0], den);
            dir[1] = div_rn(pt[1], den);
            dir[2] = div_rn(pt[2], den);
            }
            wout[(size_t)row * J + j] = w;
        }
        }
        __syncthreads();
        // coalesced copy-out of the finished rows
        const int base = tile * AN_TS * SUB;
        const int live = min(AN_TS * SUB, nrows - base) * in_ch;
        float* dst = x0 + (size_t)base * in_ch;
        if (COMPACT) {
            for (int i = tid; i < live / 4; i += AN_BLOCK) reinterpret_cast<float4*>(dst)[i] = reinterpret_cast<const float4*>(s_row)[i];
        } else {
            for (int i = tid; i < live; i += AN_BLOCK) dst[i] = s_row[i];
        }
    }
}

// reference: transform_batch_rays (encoders.py:305-317) -> VecNormEncoder -> the frequency part of
// CutoffEmbedder._embed with dist_inputs (cutoff_embedder.py:156-166); the cutoff weight is applied per
// sample in k_anerf_color.  E[ray][b*72 + 3j + k], b = 0: d, 1+2l: sin(2^l d), 2+2l: cos(2^l d)
__global__ __launch_bounds__(256) void k_anerf_view_pe(const float* __restrict__ rays_d, const float* __restrict__ skts,
                                                       int R, int G, int L, float* __restrict__ E) {
    const int rays_per_pose = R / G;
    const int stride = (1 + 2 * L) * 3 * J;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (long)R * J; i += (long)gridDim.x * blockDim.x) {
        const int r = (int)(i / J), j = (int)(i % J);
        const float* M = skts + ((size_t)min(r / rays_per_pose, G - 1) * J + j) * 16;
        const float d[3] = {rays_d[3 * r], rays_d[3 * r + 1], rays_d[3 * r + 2]};
        float q[3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
            q[a] = add_rn(add_rn(mul_rn(M[4 * a], d[0]), mul_rn(M[4 * a + 1], d[1])), mul_rn(M[4 * a + 2], d[2]));
        const float nrm = norm3_torch(q[0], q[1], q[2]);
        const float den = fmaxf(nrm, 1e-12f);
        float* out = E + (size_t)r * stride + 3 * j;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float u = div_rn(q[k], den);
            out[k] = u;
            for (int l = 0; l < L; ++l) {
                float sn, cs;
                sincosf(mul_rn(u, (float)(1 << l)), &sn, &cs);
                out[(1 + 2 * l) * 3 * J + k] = sn;
                out[(2 + 2 * l) * 3 * J + k] = cs;
            }
        }
    }
}

// reference: the view branch of NeRF.inference (nerf.py:196-209) on encode_views' output (nerf.py:252-279).
// One wavefront per ray: the ray's 24 x VW joint vectors stay in registers while its S samples stream by.
//   x[c]   = relu(featv[row][c] + table[cam][c] + sum_j w[row][j] * C[j][ray][c])
//   raw    = (rgb_w x + rgb_b, alpha[row])
constexpr int AN_VW_MAX = 256;  // 4 columns per lane
__global__ __launch_bounds__(256) void k_anerf_color(const float* __restrict__ featv, int ldf, const float* __restrict__ w,
                                                     const float* __restrict__ C, const float* __restrict__ table,
                                                     const int64_t* __restrict__ cam_idx, int n_codes, int R_total,
                                                     int ray0, int nrays, int S, int VW,
                                                     const float* __restrict__ rgb_w, const float* __restrict__ rgb_b,
                                                     const float* __restrict__ alpha, int lda, float* __restrict__ raw_out) {
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
    for (int rl = wave_global; rl < nrays; rl += nwaves) {
        const int ray = ray0 + rl;
        float cj[J][4];
#pragma unroll
        for (int j = 0; j < J; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = lane + 64 * i;
                cj[j][i] = c < VW ? C[((size_t)j * R_total + ray) * VW + c] : 0.f;
            }
        long code = n_codes;  // the mean code (Optcodes eval with idx < 0)
        if (cam_idx) {
            const long idx = cam_idx[ray];
            if (idx >= 0) code = idx < n_codes ? idx : n_codes - 1;
        }
        float tb[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = lane + 64 * i;
            tb[i] = c < VW ? table[(size_t)code * VW + c] : 0.f;
        }
        // four samples per trip: all their loads (16 feature values and 4 x 24 cutoff weights) are issued before the first
        // FMA, so one memory latency is paid per four samples instead of per sample (a wavefront walks its ray alone)
        constexpr int U = 4;
        for (int s0 = 0; s0 < S; s0 += U) {
            float x[U][4], wj[U][J], al[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const size_t row = (size_t)rl * S + min(s0 + u, S - 1);
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
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float xr = fmaxf(x[u][i], 0.f);
                    pr = fmaf(xr, rw[0][i], pr);
                    pg = fmaf(xr, rw[1][i], pg);
                    pb = fmaf(xr, rw[2][i], pb);
                }
                pr = wave_total(pr); pg = wave_total(pg); pb = wave_total(pb);   // DPP scan: no LDS-crossbar shuffles
                if (lane == 0 && s0 + u < S)
                    reinterpret_cast<float4*>(raw_out)[(size_t)ray * S + s0 + u] = make_float4(pr + rb0, pg + rb1, pb + rb2, al[u]);
            }
        }
    }
}

}  // namespace danbo

using namespace danbo;

static int anerf_encode_impl(const float* rays_o, const float* rays_d, const float* z, const float* pts, int R,
                             int S, int G, const float* skts, const float* align, const float* cutoff, float tau, const float* tau_dev,
                             int L, long row0, int nrows, float* x0, float* w_out, void* stream) {
    DANBO_CHECK_ARG(R > 0 && S > 0 && G > 0 && R % G == 0 && L >= 0 && L <= AN_MAXL && nrows >= 0 && row0 >= 0);
    DANBO_CHECK_ARG(row0 + nrows <= (long)R * S && skts && align && cutoff && x0 && w_out);
    DANBO_CHECK_ARG((z == nullptr) != (pts == nullptr) && (pts || (rays_o && rays_d)));
    if (nrows == 0) return 0;
    const int in_ch = (1 + 2 * L) * J + 3 * J;
    const int ntiles = ceil_div(nrows, AN_TS);
    const int grid = ntiles < num_cu() * 8 ? ntiles : num_cu() * 8;
    hipLaunchKernelGGL(k_anerf_encode<false>, dim3(grid), dim3(AN_BLOCK), AN_TS * in_ch * sizeof(float), (hipStream_t)stream,
                       rays_o, rays_d, z, pts, R, S, G, skts, align, cutoff, tau, L, row0, nrows, x0, w_out, tau_dev);
    DANBO_LAUNCH_RET();
}

extern "C" int danbo_anerf_encode_fwd(const float* rays_o, const float* rays_d, const float* z, const float* pts, int R,
                                 int S, int G, const float* skts, const float* align, const float* cutoff, float tau,
                                 int L, long row0, int nrows, float* x0, float* w_out, void* stream) {
    return anerf_encode_impl(rays_o, rays_d, z, pts, R, S, G, skts, align, cutoff, tau, nullptr, L, row0, nrows, x0, w_out, stream);
}

/* the same with tau read from DEVICE memory when the kernel runs (a launch inside a captured graph: the training step) */
extern "C" int danbo_anerf_encode_fwd_dtau(const float* rays_o, const float* rays_d, const float* z, const float* pts, int R,
                                 int S, int G, const float* skts, const float* align, const float* cutoff, const float* tau_dev,
                                 int L, long row0, int nrows, float* x0, float* w_out, void* stream) {
    DANBO_CHECK_ARG(tau_dev != nullptr);
    return anerf_encode_impl(rays_o, rays_d, z, pts, R, S, G, skts, align, cutoff, 0.f, tau_dev, L, row0, nrows, x0, w_out, stream);
}

/* the encoder's inputs instead of its output: table [nrows, 144] for danbo_linear16_fwd_enc (k_linear16.hip) + the cutoff weights */
extern "C" int danbo_anerf_encode_compact(const float* rays_o, const float* rays_d, const float* z, const float* pts, int R,
                                          int S, int G, const float* skts, const float* align, const float* cutoff, float tau,
                                          long row0, int nrows, float* table, float* w_out, void* stream) {
    DANBO_CHECK_ARG(R > 0 && S > 0 && G > 0 && R % G == 0 && nrows >= 0 && row0 >= 0);
    DANBO_CHECK_ARG(row0 + nrows <= (long)R * S && skts && align && cutoff && table && w_out && (uintptr_t)table % 16 == 0);
    DANBO_CHECK_ARG((z == nullptr) != (pts == nullptr) && (pts || (rays_o && rays_d)));
    if (nrows == 0) return 0;
    const int ntiles = ceil_div(nrows, AN_TS * AN_SUB);
    const int grid = ntiles < num_cu() * 8 ? ntiles : num_cu() * 8;
    hipLaunchKernelGGL(k_anerf_encode<true>, dim3(grid), dim3(AN_BLOCK), AN_SUB * AN_TS * AN_ENC_FLOATS * sizeof(float), (hipStream_t)stream,
                       rays_o, rays_d, z, pts, R, S, G, skts, align, cutoff, tau, 0, row0, nrows, table, w_out);
    DANBO_LAUNCH_RET();
}

extern "C" int danbo_anerf_view_pe_fwd(const float* rays_d, const float* skts, int R, int G, int L, float* E, void* stream) {
    DANBO_CHECK_ARG(R > 0 && G > 0 && R % G == 0 && L >= 0 && L <= AN_MAXL && rays_d && skts && E);
    hipLaunchKernelGGL(k_anerf_view_pe, dim3(stream_grid((long)R * J, 256)), dim3(256), 0, (hipStream_t)stream, rays_d,
                       skts, R, G, L, E);
    DANBO_LAUNCH_RET();
}

extern "C" int danbo_anerf_color_fwd(const float* featv, int ld_featv, const float* w, const float* C, const float* table,
                                const int64_t* cam_idx, int n_codes, int R_total, int ray0, int nrays, int S, int VW,
                                const float* rgb_w, const float* rgb_b, const float* alpha, int ld_alpha, float* raw_out,
                                void* stream) {
    DANBO_CHECK_ARG(featv && w && C && table && rgb_w && rgb_b && alpha && raw_out && ld_featv >= VW && ld_alpha >= 1);
    DANBO_CHECK_ARG(VW > 0 && VW <= AN_VW_MAX && S > 0 && nrays >= 0 && ray0 >= 0 && ray0 + nrays <= R_total && n_codes >= 0);
    if (nrays == 0) return 0;
    const int blocks = ceil_div(nrays, 4);
    const int grid = blocks < num_cu() * 8 ? blocks : num_cu() * 8;
    hipLaunchKernelGGL(k_anerf_color, dim3(grid), dim3(256), 0, (hipStream_t)stream, featv, ld_featv, w, C, table, cam_idx,
                       n_codes, R_total, ray0, nrays, S, VW, rgb_w, rgb_b, alpha, ld_alpha, raw_out);
    DANBO_LAUNCH_RET();
}
