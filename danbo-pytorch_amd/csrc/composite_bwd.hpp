// K4 backward of ONE ray by one wavefront (S <= 256), shared by k_composite_bwd (k_backward.hip) and the training step's fused
// k_train_mid (k_train_rows.hip).  gfx950 only.
//   w_i = a_i T_i, T_i = prod_{k<i} (1 - a_k + 1e-10), rgb_map = sum w_i c_i, acc = min(sum w_i, 1)
//   dL/dw_i = <g_rgb, c_i> + g_acc [sum w < 1]
//   dL/da_i = dL/dw_i T_i - (sum_{k>i} dL/dw_k w_k) / (1 - a_i + 1e-10)
//   a_i = 1 - exp(-s_i delta_i), s_i = relu(raw3_i / B + noise_i)
// out(s, d_raw of sample s) is called by the lane that owns sample s = 64 c + lane.
#pragma once
#include "common.hpp"

namespace danbo {

// per-lane state of one ray's composite between its two sweeps (chunk c holds sample 64 c + lane)
// NC: 64-sample chunks the ray can have (S <= 64 NC); k_train_mid instantiates what the launch needs -- with NC = 4 two states do not
// fit the 128 registers of a 16-wavefront workgroup
template <int NC>
struct CompositeState {
    float al[NC], T[NC], dist[NC], sig[NC], cr[NC], cg[NC], cb[NC], rr[NC], rg[NC], rb[NC];
    float acc;
    bool inside[NC];     // lazily filled raw: the sample lies in >= 1 volume (true where there are no bits)
};

// forward sweep: needs nothing of the upstream gradient (k_train_mid runs it while the loss inputs are still in flight)
template <int NC>
__device__ __forceinline__ void composite_fwd_sweep(const float4* __restrict__ raw, const float* __restrict__ z, const float* __restrict__ rays_d,
                                                    int r, int S, float B, const float* __restrict__ noise,
                                                    const float4* __restrict__ raw_empty, const uint32_t* __restrict__ bits, int lane,
                                                    CompositeState<NC>& st) {
    const int nchunk = (S + 63) >> 6;
    const float dx = rays_d[3 * r], dy = rays_d[3 * r + 1], dz_ = rays_d[3 * r + 2];
    const float dn = norm3_torch(dx, dy, dz_);
    float (&al)[NC] = st.al, (&T)[NC] = st.T, (&dist)[NC] = st.dist, (&sig)[NC] = st.sig, (&cr)[NC] = st.cr, (&cg)[NC] = st.cg, (&cb)[NC] = st.cb;
    float (&rr)[NC] = st.rr, (&rg)[NC] = st.rg, (&rb)[NC] = st.rb;
    float carry = 1.0f, acc = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        al[c] = 0.f; T[c] = 0.f; dist[c] = 0.f; sig[c] = 0.f; cr[c] = cg[c] = cb[c] = 0.f; rr[c] = rg[c] = rb[c] = 0.f;
        st.inside[c] = false;
        if (c >= nchunk) continue;
        const int s = c * 64 + lane;
        const bool act = s < S;
        const size_t m = (size_t)r * S + (act ? s : S - 1);
        // lazily filled raw: samples outside every volume were never written and take the ray's empty-space raw
        const bool in = bits == nullptr || bits[m] != 0u;
        st.inside[c] = in;
        const float4 rw = in ? raw[m] : raw_empty[r];
        const float zs = z[m];
        const float zn = (s + 1 < S) ? z[m + 1] : zs;
        dist[c] = mul_rn((s + 1 < S) ? sub_rn(zn, zs) : 1e10f, dn);
        float sg = div_rn(rw.w, B);
        if (noise) sg = add_rn(sg, noise[m]);
        sig[c] = act ? sg : -1.f;
        const float a = act ? sub_rn(1.0f, expf(-mul_rn(fmaxf(sg, 0.f), dist[c]))) : 0.f;
        al[c] = a;
        float p = act ? add_rn(sub_rn(1.0f, a), 1e-10f) : 1.0f;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const float q = __shfl_up(p, off, 64);
            if (lane >= off) p = mul_rn(p, q);
        }
        float excl = __shfl_up(p, 1, 64);
        if (lane == 0) excl = 1.0f;
        T[c] = mul_rn(carry, excl);
        carry = mul_rn(carry, __shfl(p, 63, 64));
        rr[c] = sigmoidf_(rw.x); rg[c] = sigmoidf_(rw.y); rb[c] = sigmoidf_(rw.z);
        cr[c] = rr[c] * 1.002f - 0.001f; cg[c] = rg[c] * 1.002f - 0.001f; cb[c] = rb[c] * 1.002f - 0.001f;
        acc += wave_sum(act ? a * T[c] : 0.f);
    }
    st.acc = acc;
}

// backward sweep: suffix sums of dL/dw_k * w_k, chunks in reverse
template <int NC, class Out>
__device__ __forceinline__ void composite_bwd_sweep(const CompositeState<NC>& st, int S, float B, float gr, float gg, float gb, float g_acc_r, int lane,
                                                    const Out& out) {
    const int nchunk = (S + 63) >> 6;
    const float (&al)[NC] = st.al, (&T)[NC] = st.T, (&dist)[NC] = st.dist, (&sig)[NC] = st.sig, (&cr)[NC] = st.cr, (&cg)[NC] = st.cg, (&cb)[NC] = st.cb;
    const float (&rr)[NC] = st.rr, (&rg)[NC] = st.rg, (&rb)[NC] = st.rb;
    const float ga = st.acc < 1.0f ? g_acc_r : 0.f;
    float tail = 0.f;
#pragma unroll
    for (int c = NC - 1; c >= 0; --c) {
        if (c >= nchunk) continue;
        const int s = c * 64 + lane;
        const bool act = s < S;
        const float w = al[c] * T[c];
        const float dLdw = gr * cr[c] + gg * cg[c] + gb * cb[c] + ga;
        float v = act ? dLdw * w : 0.f;
        // inclusive suffix scan inside the chunk
        float suf = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const float q = __shfl_down(suf, off, 64);
            if (lane + off < 64) suf += q;
        }
        const float after = suf - v + tail;  // strictly later samples
        tail += __shfl(suf, 0, 64);
        if (act) {
            const float dLda = dLdw * T[c] - after / (1.0f - al[c] + 1e-10f);
            const float dads = sig[c] > 0.f ? dist[c] * expf(-sig[c] * dist[c]) : 0.f;
            float4 o;
            o.x = gr * w * 1.002f * rr[c] * (1.0f - rr[c]);
            o.y = gg * w * 1.002f * rg[c] * (1.0f - rg[c]);
            o.z = gb * w * 1.002f * rb[c] * (1.0f - rb[c]);
            o.w = dLda * dads / B;
            out(s, o);
        }
    }
}

template <class Out>
__device__ __forceinline__ void composite_bwd_ray(const float4* __restrict__ raw, const float* __restrict__ z, const float* __restrict__ rays_d,
                                                  int r, int S, float B, const float* __restrict__ noise, float gr, float gg, float gb,
                                                  float g_acc_r, const float4* __restrict__ raw_empty, const uint32_t* __restrict__ bits,
                                                  int lane, const Out& out) {
    CompositeState<4> st;
    composite_fwd_sweep(raw, z, rays_d, r, S, B, noise, raw_empty, bits, lane, st);
    composite_bwd_sweep(st, S, B, gr, gg, gb, g_acc_r, lane, out);
}

}  // namespace danbo
