// Pose stage: axis-angle -> rot6d -> PE -> skeleton GNN -> per-bone factorised volumes.
// Runs once per distinct pose (G per call; G = 1 when rendering), 1.7 MMAC per pose: launch-
// and weight-streaming-bound, so each layer is one launch of G*24 workgroups (one bone of one
// pose each) that stream that bone's [Cin x Cout] weight slice with coalesced loads.  The
// 24x24 skeleton-adjacency mix of a GCN layer is folded into the NEXT layer's input stage.
// gfx950 only.
#include "common.hpp"

namespace danbo {

// pytorch3d.transforms.axis_angle_to_matrix (via quaternion; Taylor branch below 1e-6 rad);
// rot6d = first two columns, row-major (core/utils/skeleton_utils.py:408-418)
__device__ __forceinline__ void axis_angle_to_rot6d(const float* aa, float* r6) {
    const float ang = norm3_torch(aa[0], aa[1], aa[2]);
    const float half = mul_rn(ang, 0.5f);
    const float s = fabsf(ang) < 1e-6f ? sub_rn(0.5f, div_rn(mul_rn(ang, ang), 48.0f)) : div_rn(sinf(half), ang);
    const float qr = cosf(half), qi = mul_rn(aa[0], s), qj = mul_rn(aa[1], s), qk = mul_rn(aa[2], s);
    const float two_s = div_rn(2.0f, add_rn(add_rn(add_rn(mul_rn(qr, qr), mul_rn(qi, qi)), mul_rn(qj, qj)), mul_rn(qk, qk)));
    r6[0] = 1.0f - two_s * (qj * qj + qk * qk);
    r6[1] = two_s * (qi * qj - qk * qr);
    r6[2] = two_s * (qi * qj + qk * qr);
    r6[3] = 1.0f - two_s * (qi * qi + qk * qk);
    r6[4] = two_s * (qi * qk - qj * qr);
    r6[5] = two_s * (qj * qk + qi * qr);
}

// MODE 0: x = PE(rot6d(bones[g][j])), root zeroed            -> y = x W[j]
// MODE 1: x = relu(scale * (sum_j' A[j][j'] Yp[g][j'] + bp))  -> y = x W[j]   (+ bias if given)
// MODE 2: x = relu(Yp[g][j])                                  -> y = x W[j] + bias[j]
template <int MODE>
__global__ __launch_bounds__(256) void k_pose_layer(const float* __restrict__ bones, int L_graph,
                                                    const float* __restrict__ Yp, const float* __restrict__ adjw_p,
                                                    const float* __restrict__ bias_p, float scale, int Cin, int Cout,
                                                    const float* __restrict__ W, const float* __restrict__ bias,
                                                    float* __restrict__ Y) {
    __shared__ float s_x[256];
    const int g = blockIdx.x / J, j = blockIdx.x % J;
    const int tid = threadIdx.x;
    if (MODE == 0) {
        if (tid < 6) {
            float r6[6];
            axis_angle_to_rot6d(bones + ((size_t)g * J + j) * 3, r6);
            const float v = (j == 0) ? 0.f : r6[tid];  // mask_root (gnn_backbone.py:687-688)
            s_x[tid] = v;
            for (int l = 0; l < L_graph; ++l) {
                float sn, cs;
                sincosf(j == 0 ? 0.f : mul_rn(r6[tid], (float)(1 << l)), &sn, &cs);
                s_x[6 * (1 + 2 * l) + tid] = (j == 0) ? 0.f : sn;
                s_x[6 * (2 + 2 * l) + tid] = (j == 0) ? 0.f : cs;
            }
        }
    } else if (MODE == 1) {
        for (int k = tid; k < Cin; k += 256) {
            // all 24 rows requested at once (a zero adjacency entry leaves acc as it is: a * y = +-0); walking the non-zero entries
            // one dependent load after the other cost more than the layer's arithmetic
            float y[J];
#pragma unroll
            for (int jp = 0; jp < J; ++jp) y[jp] = Yp[((size_t)g * J + jp) * Cin + k];
            float acc = 0.f;
#pragma unroll
            for (int jp = 0; jp < J; ++jp) {
                const float a = adjw_p[j * J + jp];
                acc = a != 0.f ? fmaf(a, y[jp], acc) : acc;
            }
            acc = scale * (acc + bias_p[k]);
            s_x[k] = fmaxf(acc, 0.f);
        }
    } else {
        for (int k = tid; k < Cin; k += 256) s_x[k] = fmaxf(Yp[((size_t)g * J + j) * Cin + k], 0.f);
    }
    __syncthreads();
    for (int c = tid; c < Cout; c += 256) {
        const float* w = W + (size_t)j * Cin * Cout + c;
        // same summation order as before (one fmaf chain over k), 32 weight loads in flight instead of 4: the layer is the
        // latency of Cin / unroll round trips (4 launches of ~30 us each for 1.7 MMAC)
        float acc = 0.f;
        int k = 0;
        for (; k + 32 <= Cin; k += 32) {
            float wv[32];
#pragma unroll
            for (int i = 0; i < 32; ++i) wv[i] = w[(size_t)(k + i) * Cout];
#pragma unroll
            for (int i = 0; i < 32; ++i) acc = fmaf(s_x[k + i], wv[i], acc);
        }
        for (; k < Cin; ++k) acc = fmaf(s_x[k], w[(size_t)k * Cout], acc);
        if (bias) acc += bias[j * Cout + c];
        Y[((size_t)g * J + j) * Cout + c] = acc;
    }
}

}  // namespace danbo

using namespace danbo;

extern "C" int danbo_pose_volumes_fwd(const float* bones, int G, int L_graph, int W, const float* w0,
                                       const float* adjw0, const float* b0, const float* w1, const float* adjw1,
                                       const float* b1, const float* w2, const float* b2, const float* w3,
                                       const float* b3, float* scratch, float* volumes, void* stream) {
    DANBO_CHECK_ARG(G > 0 && L_graph >= 0 && W > 0 && W <= 256 && 6 * (1 + 2 * L_graph) <= 256);
    DANBO_CHECK_ARG(bones && w0 && adjw0 && b0 && w1 && adjw1 && b1 && w2 && b2 && w3 && b3 && scratch && volumes);
    hipStream_t st = (hipStream_t)stream;
    const int Cin0 = 6 * (1 + 2 * L_graph);
    float* Y0 = scratch;
    float* Y1 = Y0 + (size_t)G * J * W;
    float* Y2 = Y1 + (size_t)G * J * W;
    const dim3 grid(G * J), block(256);
    // the pose-PE'd mask zeroes the root row of the INPUT; "first layer doubled" = scale 2 on its mix
    hipLaunchKernelGGL(k_pose_layer<0>, grid, block, 0, st, bones, L_graph, nullptr, nullptr, nullptr, 1.f, Cin0, W, w0,
                       nullptr, Y0);
    hipLaunchKernelGGL(k_pose_layer<1>, grid, block, 0, st, nullptr, 0, Y0, adjw0, b0, 2.f, W, W, w1, nullptr, Y1);
    hipLaunchKernelGGL(k_pose_layer<1>, grid, block, 0, st, nullptr, 0, Y1, adjw1, b1, 1.f, W, W, w2, b2, Y2);
    hipLaunchKernelGGL(k_pose_layer<2>, grid, block, 0, st, nullptr, 0, Y2, nullptr, nullptr, 1.f, W, DANBO_VOL, w3, b3,
                       volumes);
    DANBO_LAUNCH_RET();
}
