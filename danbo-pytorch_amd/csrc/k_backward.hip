// Backward kernels of the two non-GEMM stages that carry gradients in training:
//   * factorised gather  (d part_feat -> d volumes, d axis_scale)   [K1b backward]
//   * alpha compositing   (d rgb_map, d acc_map -> d raw)            [K4 backward]
// The linear layers of the training path are plain GEMMs and go through rocBLAS (torch.addmm /
// bmm on the compacted in-volume rows); see core/train_path.py.  gfx950 only.
#include "common.hpp"
#include "composite_bwd.hpp"

namespace danbo {

// ======================================================================================
// K1b backward.  Forward (sample_math.hpp): feat[f*3+k] = (v0*w0 + v1*w1) * win with
// w1 = iy - floor(iy), w0 = 1 - w1, iy = ((x_k + 1)*16 - 1)/2, x_k = pt_k / |s_k|.
// Gradients (reference autograd: window detached, mask not differentiable):
//   d vol[f][y0][k] += g*win*w0,  d vol[f][y1][k] += g*win*w1
//   d x_k = 8 * sum_f g*win*(v1 - v0),   d s_k = d x_k * (-x_k / |s_k|) * sign(s_k)
// ======================================================================================
constexpr int GB_TS = 16;
constexpr int GB_BLOCK = GB_TS * J;

__global__ __launch_bounds__(GB_BLOCK) void k_bone_gather_bwd(const float* __restrict__ rays_o,
                                                              const float* __restrict__ rays_d,
                                                              const float* __restrict__ z,
                                                              const float* __restrict__ pts, int R, int S, int G,
                                                              const float* __restrict__ skts,
                                                              const float* __restrict__ align,
                                                              const float* __restrict__ axis_scale,
                                                              const float* __restrict__ volumes,
                                                              const int32_t* __restrict__ list, int n,
                                                              const float* __restrict__ d_part_feat,
                                                              float* __restrict__ d_volumes,
                                                              float* __restrict__ d_axis_scale) {
    __shared__ float s_align[J * 16];
    __shared__ double s_dscale[J * 3];     // fp64: ds_add_f32 is ~20x slower than ds_add_f64 on gfx950 (tools/probe/lds_atomic.hip)
    const int tid = threadIdx.x;
    for (int i = tid; i < J * 16; i += GB_BLOCK) s_align[i] = align[i];
    for (int i = tid; i < J * 3; i += GB_BLOCK) s_dscale[i] = 0.0;
    __syncthreads();
    const long spp = (long)(R / G) * S;
    const int sl = tid / J, j = tid % J;
    const float sc[3] = {axis_scale[3 * j], axis_scale[3 * j + 1], axis_scale[3 * j + 2]};
    float dsc[3] = {0.f, 0.f, 0.f};
    const int ntiles = (n + GB_TS - 1) / GB_TS;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int row = tile * GB_TS + sl;
        if (row >= n) continue;
        const long m = list ? list[row] : row;
        const int g = (int)min(m / spp, (long)G - 1);
        float p[3], pt[3], sk[12];
        if (pts) { p[0] = pts[3 * m]; p[1] = pts[3 * m + 1]; p[2] = pts[3 * m + 2]; }
        else {
            const int r = (int)(m / S);
            const float o[3] = {rays_o[3 * r], rays_o[3 * r + 1], rays_o[3 * r + 2]};
            const float d[3] = {rays_d[3 * r], rays_d[3 * r + 1], rays_d[3 * r + 2]};
            sample_point(o, d, z[m], p);
        }
        const float* src = skts + ((size_t)g * J + j) * 16;
#pragma unroll
        for (int i = 0; i < 12; ++i) sk[i] = src[i];
        bone_local(sk, s_align + 16 * j, p, pt);
        float x[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) x[k] = div_rn(pt[k], fabsf(sc[k]));
        const float win = coord_window(x);
        if (win == 0.f) continue;  // every feature (and gradient) of this bone is exactly zero
        const float* vol = volumes + ((size_t)g * J + j) * VOL;
        float* dvol = d_volumes + ((size_t)g * J + j) * VOL;
        const float* dg = d_part_feat + ((size_t)row * J + j) * FEAT;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float iy = div_rn(sub_rn(mul_rn(add_rn(x[k], 1.0f), (float)VRES), 1.0f), 2.0f);
            const float fl = floorf(iy);
            const float w1 = sub_rn(iy, fl), w0 = sub_rn(1.0f, w1);
            const int y0 = (int)fminf(fmaxf(fl, -2.0f), (float)VRES + 1.0f), y1 = y0 + 1;
            const bool ok0 = y0 >= 0 && y0 < VRES, ok1 = y1 >= 0 && y1 < VRES;
            float dx = 0.f;
#pragma unroll
            for (int f = 0; f < VOXF; ++f) {
                const float gq = dg[f * 3 + k] * win;
                const float v0 = ok0 ? vol[f * (VRES * 3) + y0 * 3 + k] : 0.f;
                const float v1 = ok1 ? vol[f * (VRES * 3) + y1 * 3 + k] : 0.f;
                if (ok0) atomicAdd(dvol + f * (VRES * 3) + y0 * 3 + k, gq * w0);
                if (ok1) atomicAdd(dvol + f * (VRES * 3) + y1 * 3 + k, gq * w1);
                dx += gq * (v1 - v0);
            }
            dx *= 0.5f * (float)VRES;
            dsc[k] += dx * (-x[k] / fabsf(sc[k])) * (sc[k] < 0.f ? -1.f : 1.f);
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k)
        if (dsc[k] != 0.f) atomicAdd(&s_dscale[3 * j + k], (double)dsc[k]);
    __syncthreads();
    for (int i = tid; i < J * 3; i += GB_BLOCK)
        if (s_dscale[i] != 0.0) atomicAdd(d_axis_scale + i, (float)s_dscale[i]);
}

// ======================================================================================
// K4 backward, one wavefront per ray (S <= 256).
//   w_i = a_i T_i, T_i = prod_{k<i} (1 - a_k + 1e-10), rgb_map = sum w_i c_i, acc = min(sum w_i, 1)
//   dL/dw_i = <g_rgb, c_i> + g_acc [sum w < 1]
//   dL/da_i = dL/dw_i T_i - (sum_{k>i} dL/dw_k w_k) / (1 - a_i + 1e-10)
//   a_i = 1 - exp(-s_i delta_i), s_i = relu(raw3_i / B + noise_i)
// ======================================================================================
__global__ __launch_bounds__(256) void k_composite_bwd(const float4* __restrict__ raw, const float* __restrict__ z,
                                                       const float* __restrict__ rays_d, int R, int S, float B,
                                                       const float* __restrict__ noise,
                                                       const float* __restrict__ g_rgb, const float* __restrict__ g_acc,
                                                       float4* __restrict__ d_raw, const float4* __restrict__ raw_empty,
                                                       const uint32_t* __restrict__ bits) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    for (int r = wave; r < R; r += nwaves)
        composite_bwd_ray(raw, z, rays_d, r, S, B, noise, g_rgb[3 * r], g_rgb[3 * r + 1], g_rgb[3 * r + 2], g_acc[r], raw_empty, bits, lane,
                          [&](int s, const float4& o) { d_raw[(size_t)r * S + s] = o; });
}

}  // namespace danbo

using namespace danbo;

extern "C" int danbo_bone_gather_bwd(const float* rays_o, const float* rays_d, const float* z, const float* pts, int R,
                                      int S, int G, const float* skts, const float* align, const float* axis_scale,
                                      const float* volumes, const int32_t* list, int n, const float* d_part_feat,
                                      float* d_volumes, float* d_axis_scale, void* stream) {
    DANBO_CHECK_ARG(R > 0 && S > 0 && G > 0 && R % G == 0 && n >= 0 && d_part_feat && d_volumes && d_axis_scale);
    DANBO_CHECK_ARG((z == nullptr) != (pts == nullptr));
    if (n == 0) return 0;
    const int ntiles = ceil_div(n, GB_TS);
    const int grid = ntiles < num_cu() * 4 ? ntiles : num_cu() * 4;
    hipLaunchKernelGGL(k_bone_gather_bwd, dim3(grid), dim3(GB_BLOCK), 0, (hipStream_t)stream, rays_o, rays_d, z, pts, R, S,
                       G, skts, align, axis_scale, volumes, list, n, d_part_feat, d_volumes, d_axis_scale);
    DANBO_LAUNCH_RET();
}

extern "C" int danbo_composite_bwd_lazy(const float* raw, const float* raw_empty, const uint32_t* valid_bits, const float* z,
                                         const float* rays_d, int R, int S, float B, const float* noise, const float* g_rgb,
                                         const float* g_acc, float* d_raw, void* stream) {
    DANBO_CHECK_ARG(R > 0 && S > 0 && S <= 256 && B > 0.f && raw && z && rays_d && g_rgb && g_acc && d_raw);
    DANBO_CHECK_ARG(valid_bits == nullptr || raw_empty != nullptr);
    hipLaunchKernelGGL(k_composite_bwd, dim3(stream_grid((long)R * 64, 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4*>(raw), z, rays_d, R, S, B, noise, g_rgb, g_acc,
                       reinterpret_cast<float4*>(d_raw), reinterpret_cast<const float4*>(raw_empty), valid_bits);
    DANBO_LAUNCH_RET();
}

extern "C" int danbo_composite_bwd(const float* raw, const float* z, const float* rays_d, int R, int S, float B,
                                    const float* noise, const float* g_rgb, const float* g_acc, float* d_raw,
                                    void* stream) {
    return danbo_composite_bwd_lazy(raw, nullptr, nullptr, z, rays_d, R, S, B, noise, g_rgb, g_acc, d_raw, stream);
}
