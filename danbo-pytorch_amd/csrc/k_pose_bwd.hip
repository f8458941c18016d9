// Backward of the pose stage (k_pose.hip): d volumes [G,24,240] -> gradients of the skeleton GNN's parameters
// (graph_net.layers.{0,1}.{lin.weight, adj_w, bias}, layers.{2,3}.{weight, bias}: 1.73 M of the model's 2.47 M parameters).
// Reference: FactorizeGNN / BodyGNN.forward (core/networks/gnn_backbone.py:683-704), DenseWGCN.forward (:249-266),
// ParallelLinear.forward (core/networks/misc.py:174-183); gradients = loss.backward() (core/trainer.py:563-576).
// Forward (per pose g, bone j; Y_l are the tensors danbo_pose_volumes_fwd leaves in its scratch):
//   x0 = PE(rot6d(bones)) (root zeroed)           Y0 = x0 W0_j
//   x1 = relu(2 (A0 Y0 + b0))                     Y1 = x1 W1_j         (the doubled first layer, gnn_backbone.py:698-699)
//   x2 = relu(A1 Y1 + b1)                         Y2 = x2 W2_j + b2_j
//   x3 = relu(Y2)                                 vol = x3 W3_j + b3_j
// Like the forward this stage is weight-streaming bound (6.9 MB of weights read once, 6.9 MB of gradients written once,
// G <= 16 poses of arithmetic per weight): one workgroup per (bone, 16 input rows) streams its [16 x Cout] weight slice with
// coalesced loads and produces BOTH that slice's gradient (sum over poses) and the input gradient of its 16 rows.
// The adjacency mixes are small separate kernels (one workgroup per pose).  gfx950 only.
#include "common.hpp"

namespace danbo {

__device__ void axis_angle_to_rot6d_b(const float* aa, float* r6) {   // same arithmetic as k_pose.hip
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

constexpr int PB_KS = 16;   // input rows per workgroup
constexpr int PB_GC = 16;   // poses per pass

// MODE 0: x = PE(rot6d(bones[g][j])), root zeroed              (no input gradient)
// MODE 1: x = relu(scale (sum_j' A[j][j'] Yp[g][j'] + bp))      dx = gradient w.r.t. x (the mix kernel takes it from there)
// MODE 2: x = relu(Yp[g][j])                                    dx = gradient w.r.t. Yp (ReLU mask applied)
template <int MODE>
__global__ __launch_bounds__(256) void k_pose_layer_bwd(const float* __restrict__ bones, int L_graph, const float* __restrict__ Yp,
                                                        const float* __restrict__ adj_w, const float* __restrict__ adj,
                                                        const float* __restrict__ bias_p, float scale, int G, int Cin, int Cout,
                                                        const float* __restrict__ W, const float* __restrict__ dY,
                                                        float* __restrict__ gW, float* __restrict__ gbias, float* __restrict__ dx) {
    __shared__ float s_dy[PB_GC][256];
    __shared__ float s_x[PB_GC][PB_KS];
    const int j = blockIdx.x, k0 = blockIdx.y * PB_KS;
    const int tid = threadIdx.x, kl = tid >> 4, cl = tid & 15;
    const int k = k0 + kl;
    const bool k_ok = k < Cin;
    for (int gc = 0; gc < G; gc += PB_GC) {
        const int ng = min(PB_GC, G - gc);
        __syncthreads();
        for (int i = tid; i < ng * Cout; i += 256) s_dy[i / Cout][i % Cout] = dY[((size_t)(gc + i / Cout) * J + j) * Cout + i % Cout];
        // recompute the layer's input rows k0 .. k0 + 15 for these poses
        {
            const int gl = tid >> 4, kk = k0 + (tid & 15);     // pose gl, input kk
            float v = 0.f;
            if (gl < ng && kk < Cin) {
                const int g = gc + gl;
                if (MODE == 0) {
                    if (j != 0) {
                        float r6[6];
                        axis_angle_to_rot6d_b(bones + ((size_t)g * J + j) * 3, r6);
                        if (kk < 6) v = r6[kk];
                        else {
                            const int b = (kk - 6) / 6, e = (kk - 6) % 6, l = b >> 1;
                            float sn, cs;
                            sincosf(mul_rn(r6[e], (float)(1 << l)), &sn, &cs);
                            v = (b & 1) ? cs : sn;
                        }
                    }
                } else if (MODE == 1) {
                    float acc = 0.f;
                    for (int jp = 0; jp < J; ++jp) {
                        const float aw = adj_w[j * J + jp] * adj[j * J + jp];
                        if (aw != 0.f) acc = fmaf(aw, Yp[((size_t)g * J + jp) * Cin + kk], acc);
                    }
                    v = fmaxf(scale * (acc + bias_p[kk]), 0.f);
                } else {
                    v = fmaxf(Yp[((size_t)g * J + j) * Cin + kk], 0.f);
                }
            }
            s_x[gl][tid & 15] = v;
        }
        __syncthreads();
        float part[PB_GC];
#pragma unroll
        for (int g = 0; g < PB_GC; ++g) part[g] = 0.f;
        for (int c = cl; c < Cout; c += 16) {
            const float w = k_ok ? W[((size_t)j * Cin + k) * Cout + c] : 0.f;
            float gw = 0.f;
#pragma unroll
            for (int g = 0; g < PB_GC; ++g) {
                const float dy = g < ng ? s_dy[g][c] : 0.f;
                gw = fmaf(s_x[g][kl], dy, gw);
                part[g] = fmaf(dy, w, part[g]);
            }
            if (k_ok) {
                float* dst = gW + ((size_t)j * Cin + k) * Cout + c;
                *dst = gc == 0 ? gw : *dst + gw;
            }
            if (gbias != nullptr && k0 == 0 && kl == 0) {       // the first row's lanes also own the bias gradient
                float gb = 0.f;
#pragma unroll
                for (int g = 0; g < PB_GC; ++g) gb += g < ng ? s_dy[g][c] : 0.f;
                float* dst = gbias + (size_t)j * Cout + c;
                *dst = gc == 0 ? gb : *dst + gb;
            }
        }
        if (MODE != 0) {
            // the 16 lanes that share k form one DPP row: reduce their partial input gradients
#pragma unroll
            for (int g = 0; g < PB_GC; ++g) {
                float x = part[g];
                DANBO_DPP_STEP(dpp_add_, 0.f, 0x111, 0xf) DANBO_DPP_STEP(dpp_add_, 0.f, 0x112, 0xf)
                DANBO_DPP_STEP(dpp_add_, 0.f, 0x114, 0xf) DANBO_DPP_STEP(dpp_add_, 0.f, 0x118, 0xf)
                part[g] = x;
            }
            if (cl == 15 && k_ok) {
#pragma unroll
                for (int g = 0; g < PB_GC; ++g) {
                    if (g < ng) {
                        float v = part[g];
                        if (MODE == 2) v = Yp[((size_t)(gc + g) * J + j) * Cin + k] > 0.f ? v : 0.f;
                        dx[((size_t)(gc + g) * J + j) * Cin + k] = v;
                    }
                }
            }
        }
    }
}

// adjoint of  x = relu(scale (A Yp + b))  for one pose per workgroup, one thread per channel:
//   dM = [x > 0] scale dx;  dYp[j'] = sum_j A[j][j'] dM[j];  d b += sum_j dM[j];  d adj_w[j][j'] += adj[j][j'] sum_k dM[j][k] Yp[j'][k]
__global__ __launch_bounds__(256) void k_pose_mix_bwd(const float* __restrict__ dx, const float* __restrict__ Yp,
                                                      const float* __restrict__ adj_w, const float* __restrict__ adj,
                                                      const float* __restrict__ bias, float scale, int C, float* __restrict__ dYp,
                                                      float* __restrict__ g_bias, float* __restrict__ g_adj_w) {
    __shared__ float s_A[J * J], s_dA[J * J], s_edge[J * J];
    __shared__ float s_dm[J][256], s_y[J][256];     // this pose's dM and Yp, channel-minor (C <= 256)
    const int g = blockIdx.x, tid = threadIdx.x;
    for (int i = tid; i < J * J; i += 256) { s_A[i] = adj_w[i] * adj[i]; s_edge[i] = adj[i]; s_dA[i] = 0.f; }
    __syncthreads();
    for (int k = tid; k < 256; k += 256) {
        float y[J], dm[J];
        if (k >= C) {
#pragma unroll
            for (int j = 0; j < J; ++j) { s_dm[j][tid] = 0.f; s_y[j][tid] = 0.f; }
            continue;
        }
#pragma unroll
        for (int j = 0; j < J; ++j) y[j] = Yp[((size_t)g * J + j) * C + k];
        float db = 0.f;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            float acc = 0.f;
#pragma unroll
            for (int jp = 0; jp < J; ++jp) acc = fmaf(s_A[j * J + jp], y[jp], acc);
            const float m = scale * (acc + bias[k]);
            dm[j] = m > 0.f ? scale * dx[((size_t)g * J + j) * C + k] : 0.f;
            db += dm[j];
        }
#pragma unroll
        for (int jp = 0; jp < J; ++jp) {
            float acc = 0.f;
#pragma unroll
            for (int j = 0; j < J; ++j) acc = fmaf(s_A[j * J + jp], dm[j], acc);
            dYp[((size_t)g * J + jp) * C + k] = acc;
        }
        if (db != 0.f) atomicAdd(g_bias + k, db);
#pragma unroll
        for (int j = 0; j < J; ++j) { s_dm[j][tid] = dm[j]; s_y[j][tid] = y[j]; }
    }
    __syncthreads();
    // d A[j][j'] = sum_k dM[j][k] Yp[j'][k]: one wavefront per edge (lanes over channels), not 128 threads adding into the same
    // LDS word (that version spent 100 us serialising 9 000 same-address atomics)
    {
        const int lane = tid & 63, wave = tid >> 6;
        for (int e = wave; e < J * J; e += 4) {
            if (s_edge[e] == 0.f) continue;
            const int j = e / J, jp = e % J;
            float v = 0.f;
            for (int k = lane; k < C; k += 64) v += s_dm[j][k] * s_y[jp][k];
            v = wave_total(v);
            if (lane == 0) s_dA[e] = v;
        }
    }
    __syncthreads();
    for (int i = tid; i < J * J; i += 256)
        if (s_dA[i] != 0.f) atomicAdd(g_adj_w + i, s_dA[i] * s_edge[i]);
}

}  // namespace danbo

using namespace danbo;

extern "C" int danbo_pose_volumes_bwd(const float* bones, int G, int L_graph, int W, const float* w0, const float* adj_w0,
                                      const float* adj0, const float* b0, const float* w1, const float* adj_w1, const float* adj1,
                                      const float* b1, const float* w2, const float* w3, const float* fwd_scratch /*Y0,Y1,Y2*/,
                                      const float* d_volumes, float* g_w0, float* g_adj_w0, float* g_b0, float* g_w1,
                                      float* g_adj_w1, float* g_b1, float* g_w2, float* g_b2, float* g_w3, float* g_b3,
                                      float* bwd_scratch /*>= 2*G*24*W floats*/, void* stream) {
    DANBO_CHECK_ARG(G > 0 && L_graph >= 0 && W > 0 && W <= 256 && 6 * (1 + 2 * L_graph) <= 256);
    DANBO_CHECK_ARG(bones && w0 && adj_w0 && adj0 && b0 && w1 && adj_w1 && adj1 && b1 && w2 && w3 && fwd_scratch && d_volumes && bwd_scratch);
    DANBO_CHECK_ARG(g_w0 && g_adj_w0 && g_b0 && g_w1 && g_adj_w1 && g_b1 && g_w2 && g_b2 && g_w3 && g_b3);
    hipStream_t st = (hipStream_t)stream;
    const int Cin0 = 6 * (1 + 2 * L_graph);
    const float* Y0 = fwd_scratch;
    const float* Y1 = Y0 + (size_t)G * J * W;
    const float* Y2 = Y1 + (size_t)G * J * W;
    float* da = bwd_scratch;                     // gradient w.r.t. a layer's input
    float* db = da + (size_t)G * J * W;          // gradient w.r.t. the previous layer's output
    const dim3 block(256);
    auto grid = [](int cin) { return dim3(J, (cin + PB_KS - 1) / PB_KS); };
    // layer 3: x3 = relu(Y2);  da = d Y2
    hipLaunchKernelGGL(k_pose_layer_bwd<2>, grid(W), block, 0, st, nullptr, 0, Y2, nullptr, nullptr, nullptr, 1.f, G, W, DANBO_VOL, w3,
                       d_volumes, g_w3, g_b3, da);
    // layer 2: x2 = relu(A1 Y1 + b1);  db = d x2, then the mix gives da = d Y1
    hipLaunchKernelGGL(k_pose_layer_bwd<1>, grid(W), block, 0, st, nullptr, 0, Y1, adj_w1, adj1, b1, 1.f, G, W, W, w2, da, g_w2, g_b2, db);
    hipLaunchKernelGGL(k_pose_mix_bwd, dim3(G), block, 0, st, db, Y1, adj_w1, adj1, b1, 1.f, W, da, g_b1, g_adj_w1);
    // layer 1: x1 = relu(2 (A0 Y0 + b0))
    hipLaunchKernelGGL(k_pose_layer_bwd<1>, grid(W), block, 0, st, nullptr, 0, Y0, adj_w0, adj0, b0, 2.f, G, W, W, w1, da, g_w1, nullptr, db);
    hipLaunchKernelGGL(k_pose_mix_bwd, dim3(G), block, 0, st, db, Y0, adj_w0, adj0, b0, 2.f, W, da, g_b0, g_adj_w0);
    // layer 0: x0 = PE(rot6d)
    hipLaunchKernelGGL(k_pose_layer_bwd<0>, grid(Cin0), block, 0, st, bones, L_graph, nullptr, nullptr, nullptr, nullptr, 1.f, G, Cin0, W,
                       w0, da, g_w0, nullptr, nullptr);
    DANBO_LAUNCH_RET();
}
