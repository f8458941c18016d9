// Per-item arithmetic of the sampling / transform / gather / composite stages.
//
// Everything here is plain scalar C++ marked DANBO_HD so that the very same code is
//   * inlined into the gfx950 kernels (hipcc), and
//   * compiled for the host by g++ (tests/host_emu.cpp) so the CPU test-suite can check the
//     kernel bodies against the numpy oracle without a GPU.
// Operation order is part of the contract: the transform chain and the in-volume test use
// separately rounded fp32 mul/add (never FMA) so that GPU, host build and
// oracle/danbo_oracle.py agree bit-for-bit on the in-volume mask.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define DANBO_HD __host__ __device__ __forceinline__
#else
#define DANBO_HD inline
#endif

namespace danbo {

constexpr int J = 24;
constexpr int FEAT = 15;
constexpr int VOXF = 5;
constexpr int VRES = 16;
constexpr int VOL = 240;

// ---- exactly-rounded, never-contracted fp32 primitives --------------------------------
#if defined(__HIP_DEVICE_COMPILE__)
DANBO_HD float mul_rn(float a, float b) { return __fmul_rn(a, b); }
DANBO_HD float add_rn(float a, float b) { return __fadd_rn(a, b); }
DANBO_HD float sub_rn(float a, float b) { return __fsub_rn(a, b); }
DANBO_HD float div_rn(float a, float b) { return __fdiv_rn(a, b); }
DANBO_HD double dmul_rn(double a, double b) { return __dmul_rn(a, b); }
DANBO_HD double dadd_rn(double a, double b) { return __dadd_rn(a, b); }
#else
// host build is compiled with -ffp-contract=off
DANBO_HD float mul_rn(float a, float b) { volatile float r = a * b; return r; }
DANBO_HD float add_rn(float a, float b) { volatile float r = a + b; return r; }
DANBO_HD float sub_rn(float a, float b) { volatile float r = a - b; return r; }
DANBO_HD float div_rn(float a, float b) { volatile float r = a / b; return r; }
DANBO_HD double dmul_rn(double a, double b) { volatile double r = a * b; return r; }
DANBO_HD double dadd_rn(double a, double b) { volatile double r = a + b; return r; }
#endif

// torch.norm(x, dim=-1) of a 2- / 3-vector: the reduction's step is acc + x*x CONTRACTED to one fma (ATen NormTwoOps; pinned on
// the reference's tensors in the build container: sqrt(fma(z,z,fma(y,y,x*x))) equals torch's result on every one of 98 304 rows,
// the separately rounded sum on 95 % of them) -- the only place of this file where an FMA is part of the contract
DANBO_HD float norm3_torch(float x, float y, float z) { return sqrtf(fmaf(z, z, fmaf(y, y, mul_rn(x, x)))); }
DANBO_HD float norm2_torch(float x, float y) { return sqrtf(fmaf(y, y, mul_rn(x, x))); }

// torch.linspace(0,1,n)[i] in fp32 (symmetric evaluation around the midpoint)
DANBO_HD float linspace01(int i, int n) {
    if (n <= 1) return 0.f;
    const float step = div_rn(1.0f, (float)(n - 1));
    return (i < n / 2) ? mul_rn((float)i, step) : sub_rn(1.0f, mul_rn((float)(n - 1 - i), step));
}

// sample_from_lineseg, perturb = 0:  near*(1-t) + far*t
DANBO_HD float coarse_z(float nr, float fr, int s, int S) {
    const float t = linspace01(s, S);
    return add_rn(mul_rn(nr, sub_rn(1.0f, t)), mul_rn(fr, t));
}

// pts = o + d*z (two roundings), core/raycasters.py:463
DANBO_HD void sample_point(const float* o, const float* d, float z, float* p) {
    p[0] = add_rn(o[0], mul_rn(d[0], z));
    p[1] = add_rn(o[1], mul_rn(d[1], z));
    p[2] = add_rn(o[2], mul_rn(d[2], z));
}

// ((m0*x + m1*y) + m2*z) + m3 for the three rows of a row-major 4x4 (only rows 0..2 read)
DANBO_HD void affine_unfused(const float* M, const float* p, float* q) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float* r = M + 4 * k;
        float s = add_rn(mul_rn(r[0], p[0]), mul_rn(r[1], p[1]));
        s = add_rn(s, mul_rn(r[2], p[2]));
        q[k] = add_rn(s, r[3]);
    }
}

// world -> bone-local -> bone-aligned  (core/encoders.py:288-303,442-444)
DANBO_HD void bone_local(const float* skt, const float* align, const float* p, float* pt) {
    float pl[3];
    affine_unfused(skt, p, pl);
    affine_unfused(align, pl, pt);
}

// !invalid of gnn_backbone.py:808 in the division-free form  |p_k| <= |scale_k|
DANBO_HD bool in_volume(const float* pt, const float* abs_scale) {
    return !(fabsf(pt[0]) > abs_scale[0] || fabsf(pt[1]) > abs_scale[1] || fabsf(pt[2]) > abs_scale[2]);
}

// window = exp(-2 * sum x^6), gnn_backbone.py:803-804
DANBO_HD float coord_window(const float* x) {
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float x2 = mul_rn(x[k], x[k]);
        acc = add_rn(acc, mul_rn(mul_rn(x2, x2), x2));
    }
    return expf(mul_rn(-2.0f, acc));
}

// one bone's 15 windowed features from its 240-float factorised volume
// layout vol[f*48 + r*3 + axis]; out[f*3 + axis]   (misc.py:331-351, gnn_backbone.py:810-826)
template <typename VolPtr>
DANBO_HD void gather_bone_features(VolPtr vol, const float* pt, const float* abs_scale, float* out) {
    float x[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) x[k] = div_rn(pt[k], abs_scale[k]);
    const float win = coord_window(x);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float iy = div_rn(sub_rn(mul_rn(add_rn(x[k], 1.0f), (float)VRES), 1.0f), 2.0f);
        const float fl = floorf(iy);
        const float w1 = sub_rn(iy, fl);
        const float w0 = sub_rn(1.0f, w1);
        // clamp before the int conversion: far-away samples have |iy| ~ 1e3..1e6
        const float flc = fminf(fmaxf(fl, -2.0f), (float)VRES + 1.0f);
        const int y0 = (int)flc;
        const int y1 = y0 + 1;
        const bool ok0 = (y0 >= 0) && (y0 < VRES);
        const bool ok1 = (y1 >= 0) && (y1 < VRES);
        const int c0 = ok0 ? y0 : 0;
        const int c1 = ok1 ? y1 : 0;
#pragma unroll
        for (int f = 0; f < VOXF; ++f) {
            const float v0 = ok0 ? vol[f * (VRES * 3) + c0 * 3 + k] : 0.f;
            const float v1 = ok1 ? vol[f * (VRES * 3) + c1 * 3 + k] : 0.f;
            const float s = add_rn(mul_rn(v0, w0), mul_rn(v1, w1));
            out[f * 3 + k] = mul_rn(s, win);
        }
    }
}

DANBO_HD float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// ---- get_near_far_in_cylinder, one ray, before the NaN back-fill ----------------------
// core/utils/ray_utils.py:294-329.  Returns false when the ray misses the cylinder.
DANBO_HD bool cylinder_bounds(const float* o, const float* d, const float* cyl, float near0, float far0,
                              float* nr, float* fr) {
    const float nx = add_rn(o[0], mul_rn(d[0], near0)), nz = add_rn(o[2], mul_rn(d[2], near0));
    const float fx = add_rn(o[0], mul_rn(d[0], far0)), fz = add_rn(o[2], mul_rn(d[2], far0));
    const float ncx = sub_rn(cyl[0], nx), ncz = sub_rn(cyl[1], nz);
    const float nfx = sub_rn(fx, nx), nfz = sub_rn(fz, nz);
    const float nf_norm = norm2_torch(nfx, nfz);
    const float scale = norm2_torch(d[0], d[2]);
    const float cross = sub_rn(mul_rn(ncx, nfz), mul_rn(ncz, nfx));
    const float dist = div_rn(fabsf(cross), nf_norm);
    const float rad = cyl[2];
    const float q2 = sub_rn(mul_rn(rad, rad), mul_rn(dist, dist));
    const float Q = sqrtf(q2);  // NaN when the ray misses
    const float K = div_rn(add_rn(mul_rn(ncx, nfx), mul_rn(ncz, nfz)), nf_norm);
    const float mask = (Q < K) ? 1.0f : 0.0f;
    *nr = add_rn(near0, div_rn(mul_rn(mask, sub_rn(K, Q)), scale));
    *fr = add_rn(near0, div_rn(add_rn(K, Q), scale));
    return !(Q != Q);
}

// ---- per-bone box near/far (fast configs), one ray x one bone --------------------------
// core/raycasters.py:659-697 + core/utils/ray_utils.py:383-417.  Returns true when exactly
// two of the six plane hits are inside the (bound + eps) box; *smin/*smax = their steps.
DANBO_HD bool bone_box_steps(const float* skt, const float* align, const float* abs_scale,
                             const float* o, const float* d, float* smin, float* smax) {
    const float bound = 1.3f, eps = 1e-4f;
    float ol[3], ot[3], dl[3], dt[3];
    // origin: full affine; direction: rotation only.  (torch.matmul order: sequential k.)
    for (int k = 0; k < 3; ++k) {
        const float* r = skt + 4 * k;
        ol[k] = add_rn(add_rn(add_rn(mul_rn(r[0], o[0]), mul_rn(r[1], o[1])), mul_rn(r[2], o[2])), r[3]);
        dl[k] = add_rn(add_rn(mul_rn(r[0], d[0]), mul_rn(r[1], d[1])), mul_rn(r[2], d[2]));
    }
    for (int k = 0; k < 3; ++k) {
        const float* r = align + 4 * k;
        ot[k] = add_rn(add_rn(add_rn(mul_rn(r[0], ol[0]), mul_rn(r[1], ol[1])), mul_rn(r[2], ol[2])), r[3]);
        dt[k] = add_rn(add_rn(mul_rn(r[0], dl[0]), mul_rn(r[1], dl[1])), mul_rn(r[2], dl[2]));
    }
    float os[3], ds[3];
    for (int k = 0; k < 3; ++k) { os[k] = div_rn(ot[k], abs_scale[k]); ds[k] = div_rn(dt[k], abs_scale[k]); }
    const float dnorm = norm3_torch(dt[0], dt[1], dt[2]);
    int hits = 0;
    float lo = INFINITY, hi = -INFINITY;
    const float lim = bound + eps;
    for (int side = 0; side < 2; ++side) {
        for (int ax = 0; ax < 3; ++ax) {
            const double b = side ? (double)bound : -(double)bound;
            const double t = (b - (double)os[ax]) / (double)ds[ax];
            float p[3];
            bool ok = true;
            for (int k = 0; k < 3; ++k) {
                p[k] = (float)dadd_rn(dmul_rn(t, (double)ds[k]), (double)os[k]);   // torch: t * d + o as two float64 ops
                ok = ok && (p[k] <= lim) && (p[k] >= -lim);
            }
            if (ok) {
                ++hits;
                float df[3];
                for (int k = 0; k < 3; ++k) df[k] = sub_rn(mul_rn(p[k], abs_scale[k]), ot[k]);
                const float step = div_rn(norm3_torch(df[0], df[1], df[2]), dnorm);
                lo = fminf(lo, step);
                hi = fmaxf(hi, step);
            }
        }
    }
    *smin = lo;
    *smax = hi;
    return hits == 2;
}

// ---- importance sampling of one ray (is_only pdf + inverse CDF + stable merge) --------
// core/utils/ray_utils.py:159-203,257-291.  z, w: [S]; u: [Sf] or NULL (linspace);
// scratch cdf: [S-1] floats... caller provides `cdf` with room for S floats.
DANBO_HD void importance_ray(const float* z, const float* w, int S, int Sf, const float* u, float* cdf,
                             float* z_fine, float* z_sorted, int32_t* sorted_idx) {
    const int nb = S - 2;  // number of pdf bins; bin edges are the S-1 midpoints
    float sum = 0.f;
    for (int i = 0; i < nb; ++i) {
        const float dw = add_rn(add_rn(mul_rn(0.5f, add_rn(fmaxf(w[i], w[i + 1]), fmaxf(w[i + 1], w[i + 2]))), 0.01f), 1e-5f);
        cdf[i + 1] = dw;
        sum = add_rn(sum, dw);
    }
    cdf[0] = 0.f;
    float run = 0.f;
    for (int i = 0; i < nb; ++i) {
        run = add_rn(run, div_rn(cdf[i + 1], sum));
        cdf[i + 1] = run;
    }
    const int ncdf = nb + 1;  // == S-1 == number of bin edges
    for (int k = 0; k < Sf; ++k) {
        const float uk = u ? u[k] : linspace01(k, Sf);
        // searchsorted(cdf, u, right=True): first index with cdf[idx] > u
        int lo_i = 0, hi_i = ncdf;
        while (lo_i < hi_i) {
            const int mid = (lo_i + hi_i) >> 1;
            if (cdf[mid] > uk) hi_i = mid; else lo_i = mid + 1;
        }
        const int below = lo_i - 1 > 0 ? lo_i - 1 : 0;
        const int above = lo_i < ncdf - 1 ? lo_i : ncdf - 1;
        const float c0 = cdf[below], c1 = cdf[above];
        const float b0 = mul_rn(0.5f, add_rn(z[below + 1], z[below]));
        const float b1 = mul_rn(0.5f, add_rn(z[above + 1], z[above]));
        float denom = sub_rn(c1, c0);
        if (denom < 1e-5f) denom = 1.0f;
        const float t = div_rn(sub_rn(uk, c0), denom);
        z_fine[k] = add_rn(b0, mul_rn(t, sub_rn(b1, b0)));
    }
    // stable merge, coarse first on ties (torch.sort of cat([z, z_fine]))
    int a = 0, b = 0;
    for (int i = 0; i < S + Sf; ++i) {
        const bool take_a = (b >= Sf) || (a < S && z[a] <= z_fine[b]);
        if (take_a) { z_sorted[i] = z[a]; sorted_idx[i] = a; ++a; }
        else { z_sorted[i] = z_fine[b]; sorted_idx[i] = S + b; ++b; }
    }
}

// ---- alpha compositing of one ray, sequential form (core/networks/nerf.py:281-347) -----
DANBO_HD void composite_ray(const float* raw, const float* z, const float* d, int S, float B, const float* noise,
                            float* rgb_map, float* disp, float* acc_out, float* weights, float* alpha_out) {
    const float dn = norm3_torch(d[0], d[1], d[2]);
    float T = 1.0f, r = 0.f, g = 0.f, b = 0.f, depth = 0.f, acc = 0.f;
    for (int s = 0; s < S; ++s) {
        const float dz = (s + 1 < S) ? sub_rn(z[s + 1], z[s]) : 1e10f;
        const float dist = mul_rn(dz, dn);
        float sg = div_rn(raw[4 * s + 3], B);
        if (noise) sg = add_rn(sg, noise[s]);
        sg = fmaxf(sg, 0.f);
        const float al = sub_rn(1.0f, expf(-mul_rn(sg, dist)));
        const float w = mul_rn(al, T);
        T = mul_rn(T, add_rn(sub_rn(1.0f, al), 1e-10f));
        const float cr = sub_rn(mul_rn(sigmoidf_(raw[4 * s + 0]), 1.002f), 0.001f);
        const float cg = sub_rn(mul_rn(sigmoidf_(raw[4 * s + 1]), 1.002f), 0.001f);
        const float cb = sub_rn(mul_rn(sigmoidf_(raw[4 * s + 2]), 1.002f), 0.001f);
        r = add_rn(r, mul_rn(w, cr));
        g = add_rn(g, mul_rn(w, cg));
        b = add_rn(b, mul_rn(w, cb));
        depth = add_rn(depth, mul_rn(w, z[s]));
        acc = add_rn(acc, w);
        if (weights) weights[s] = w;
        if (alpha_out) alpha_out[s] = al;
    }
    rgb_map[0] = r; rgb_map[1] = g; rgb_map[2] = b;
    float dsp = div_rn(1.0f, fmaxf(1e-10f, div_rn(depth, add_rn(acc, 1e-10f))));
    // torch.isclose(acc, 0): |acc| <= 1e-8 + 1e-5*0
    if (fabsf(acc) <= 1e-8f) dsp = 0.f;
    *disp = dsp;
    *acc_out = fminf(acc, 1.0f);
}

}  // namespace danbo
