// The reference's eager encoder helpers as kernels of their own -- for callers that use them OUTSIDE the fused path (inside it
// the same arithmetic lives in k_bone_cull / k_assign16 / k_view_consts / k_anerf_encode):
//   danbo_transform_batch_pts   core/encoders.py:288-303 transform_batch_pts  (points into every joint's local frame)
//                               core/encoders.py:305-318 transform_batch_rays (rot_only: the rotational part applied to directions)
//   danbo_optcodes_fwd          core/networks/embedding.py:17-39 Optcodes.forward (row lookup with clamping, the mean code of an
//                               evaluation without a frame, the interpolation of two codes)
// HBM-bound byte work: one thread per output vector, inputs read once, coalesced stores.  gfx950 only.
#include "common.hpp"

namespace danbo {

// out[r, s, j, :] = skt[g, j, :3, :3] p[r, s, :] (+ skt[g, j, :3, 3]), g = r / rays_per_pose; the products of a row added in the
// order k = 0, 1, 2, (translation), each rounded: a [4 x 4] @ [4 x S] product evaluated as a dot-product chain
__global__ __launch_bounds__(256) void k_transform_batch(const float* __restrict__ pts, const float* __restrict__ skt, long n_vec, int S,
                                                         int J, int rays_per_pose, int rot_only, float* __restrict__ out) {
    const long total = n_vec * J;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long v = i / J;                  // (ray, sample)
        const int j = (int)(i - v * J);
        const long ray = v / S;
        const float* m = skt + ((ray / rays_per_pose) * J + j) * 16;
        const float x = pts[3 * v], y = pts[3 * v + 1], z = pts[3 * v + 2];
        float o[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            float a = m[4 * r] * x;
            a = fmaf(m[4 * r + 1], y, a);
            a = fmaf(m[4 * r + 2], z, a);
            o[r] = rot_only ? a : a + m[4 * r + 3];
        }
        out[3 * i] = o[0]; out[3 * i + 1] = o[1]; out[3 * i + 2] = o[2];
    }
}

// mode 0: out[n] = codes[min(idx[n], n_codes - 1)] (negative indices: row 0 -- the caller has checked the range as the reference does)
// mode 1: out[n] = mean over the rows of codes (fp64 accumulation, rounded once)
// mode 2: out[n] = lerp(codes[i0[n]], codes[i1[n]], w[n]) = a + w (b - a)   (torch.lerp's formula for |w| < 0.5, else b - (b - a)(1 - w))
__global__ __launch_bounds__(256) void k_optcodes(const float* __restrict__ codes, int n_codes, int C, const float* __restrict__ idx, int idx_cols,
                                                  long N, int mode, float* __restrict__ out) {
    const long total = N * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long n = i / C;
        const int c = (int)(i - n * C);
        float v;
        if (mode == 1) {
            double acc = 0.0;
            for (int r = 0; r < n_codes; ++r) acc += (double)codes[(long)r * C + c];
            v = (float)(acc / (double)n_codes);
        } else {
            auto row = [&](float f) { const long r = (long)f; return r < 0 ? 0L : (r >= n_codes ? (long)n_codes - 1 : r); };
            if (mode == 0) v = codes[row(idx[n * idx_cols]) * C + c];
            else {
                const float a = codes[row(idx[n * idx_cols]) * C + c], b = codes[row(idx[n * idx_cols + 1]) * C + c], w = idx[n * idx_cols + 2];
                v = fabsf(w) < 0.5f ? fmaf(w, b - a, a) : b - (b - a) * (1.f - w);
            }
        }
        out[i] = v;
    }
}

}  // namespace danbo

using namespace danbo;

extern "C" int danbo_transform_batch_pts(const float* pts, const float* skt, long n_rays, int S, int J, int rays_per_pose, int rot_only,
                                          float* out, void* stream) {
    DANBO_CHECK_ARG(pts && skt && out && n_rays >= 0 && S > 0 && J > 0 && rays_per_pose > 0);
    if (n_rays == 0) return 0;
    const long n_vec = n_rays * S;
    hipLaunchKernelGGL(k_transform_batch, dim3(stream_grid(n_vec * J, 256)), dim3(256), 0, (hipStream_t)stream, pts, skt, n_vec, S, J,
                       rays_per_pose, rot_only, out);
    DANBO_LAUNCH_RET();
}

extern "C" int danbo_optcodes_fwd(const float* codes, int n_codes, int code_ch, const float* idx, int idx_cols, long N, int mode, float* out,
                                   void* stream) {
    DANBO_CHECK_ARG(codes && out && n_codes > 0 && code_ch > 0 && N >= 0 && mode >= 0 && mode <= 2);
    DANBO_CHECK_ARG(mode == 1 || (idx && idx_cols >= (mode == 2 ? 3 : 1)));
    if (N == 0) return 0;
    hipLaunchKernelGGL(k_optcodes, dim3(stream_grid(N * code_ch, 256)), dim3(256), 0, (hipStream_t)stream, codes, n_codes, code_ch, idx,
                       idx_cols, N, mode, out);
    DANBO_LAUNCH_RET();
}
