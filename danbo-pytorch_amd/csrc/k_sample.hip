// Ray bounds, sampling, world->bone transform + in-volume cull (K1a), factorised gather
// (K1b), alpha compositing (K4), importance sampling and merge.  gfx950 only.
//
// Layout notes (MI355X): samples of a pass are addressed m = r*S + s, so a wavefront's 64
// lanes read 64 consecutive z values (256 B, one coalesced request) and the 12+12 B of at most
// two rays (broadcast).  Per-pose skeleton transforms (24 x 3x4), the 24 bone-align transforms
// and |axis_scale| live in LDS for the whole workgroup; the 23 KB factorised volume of the
// pose is staged in LDS by the gather kernel.
#include "common.hpp"

namespace danbo {

// ======================================================================================
// near / far in the bounding cylinder
// ======================================================================================
__global__ __launch_bounds__(256) DANBO_NO_PK_F32 void k_cylinder_pass1(const float* __restrict__ rays_o,
                                                        const float* __restrict__ rays_d,
                                                        const float* __restrict__ cyl, int R, int G,
                                                        float near0, float far0,
                                                        const float* __restrict__ near_in,
                                                        const float* __restrict__ far_in, int chunk,
                                                        double* __restrict__ acc,  // [nchunk][4]
                                                        float* __restrict__ near_out,
                                                        float* __restrict__ far_out) {
    const int rays_per_pose = R / G;
    // wave-uniform trip count: every lane runs every iteration (inactive ones with act = false)
    for (int r0 = blockIdx.x * blockDim.x; r0 < R; r0 += gridDim.x * blockDim.x) {
        const int r = r0 + threadIdx.x;
        const bool act = r < R;
        const int rc = act ? r : R - 1;
        const int g = min(rc / rays_per_pose, G - 1);
        float o[3] = {rays_o[3 * rc], rays_o[3 * rc + 1], rays_o[3 * rc + 2]};
        float d[3] = {rays_d[3 * rc], rays_d[3 * rc + 1], rays_d[3 * rc + 2]};
        float c[3] = {cyl[5 * g], cyl[5 * g + 1], cyl[5 * g + 2]};
        float nr, fr;
        const bool hit = cylinder_bounds(o, d, c, near_in ? near_in[rc] : near0, far_in ? far_in[rc] : far0, &nr, &fr) && act;
        if (act) {
            near_out[r] = nr;
            far_out[r] = fr;
        }
        // chunk-wide sums for the nan-mean back-fill: reduce inside the wavefront first when all
        // of its rays belong to one chunk (the common case), one fp64 atomic triple per wavefront
        const int ck = rc / chunk;
        const int ck0 = __shfl(ck, 0, 64);
        const bool uniform = __all(ck == ck0);
        double sn = hit ? (double)nr : 0.0, sf = hit ? (double)fr : 0.0, sc = hit ? 1.0 : 0.0;
        if (uniform) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                sn += __shfl_xor(sn, off, 64);
                sf += __shfl_xor(sf, off, 64);
                sc += __shfl_xor(sc, off, 64);
            }
            if ((threadIdx.x & 63) == 0 && sc > 0.0) {
                double* a = acc + 4 * ck;
                atomicAdd(a + 0, sn);
                atomicAdd(a + 1, sf);
                atomicAdd(a + 2, sc);
            }
        } else if (hit) {
            double* a = acc + 4 * ck;
            atomicAdd(a + 0, sn);
            atomicAdd(a + 1, sf);
            atomicAdd(a + 2, 1.0);
        }
    }
}

__global__ __launch_bounds__(256) void k_cylinder_pass2(int R, float near0, float far0,
                                                        const float* __restrict__ near_in,
                                                        const float* __restrict__ far_in, int chunk,
                                                        const double* __restrict__ acc,
                                                        float* __restrict__ near_out,
                                                        float* __restrict__ far_out) {
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < R; r += gridDim.x * blockDim.x) {
        const float nr = near_out[r];
        if (nr != nr) {  // ray missed the cylinder: chunk-wide nan-mean (ray_utils.py:330-344)
            const double* a = acc + 4 * (r / chunk);
            const double cnt = a[2];
            near_out[r] = cnt > 0.0 ? (float)(a[0] / cnt) : (near_in ? near_in[r] : near0);
            far_out[r] = cnt > 0.0 ? (float)(a[1] / cnt) : (far_in ? far_in[r] : far0);
        }
    }
}

// Both passes in ONE launch when a chunk fits a workgroup's loop (chunk <= CYL_FUSED_MAX rays): workgroup c owns chunk c, its
// threads keep their rays' bounds, the chunk's sums are a block reduction (no atomics, no zeroed scratch, a fixed order) and the
// back-fill follows behind a barrier.  Three launches of ~5 us became one: they sit in front of everything else of a frame / step.
constexpr int CYL_BLOCK = 1024, CYL_FUSED_MAX = 16 * CYL_BLOCK;
__global__ __launch_bounds__(CYL_BLOCK) DANBO_NO_PK_F32 void k_cylinder_chunk(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                              const float* __restrict__ cyl, int R, int G, float near0, float far0,
                                                              const float* __restrict__ near_in, const float* __restrict__ far_in,
                                                              int chunk, float* __restrict__ near_out, float* __restrict__ far_out) {
    __shared__ double s_part[CYL_BLOCK / 64][3];
    __shared__ double s_sum[3];
    const int rays_per_pose = R / G;
    const int r_begin = blockIdx.x * chunk, r_end = min(r_begin + chunk, R);
    constexpr int PER = CYL_FUSED_MAX / CYL_BLOCK;
    float nr[PER], fr[PER];
    double sn = 0.0, sf = 0.0, sc = 0.0;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int r = r_begin + k * CYL_BLOCK + (int)threadIdx.x;
        nr[k] = 0.f; fr[k] = 0.f;
        if (r < r_end) {
            const int g = min(r / rays_per_pose, G - 1);
            const float o[3] = {rays_o[3 * r], rays_o[3 * r + 1], rays_o[3 * r + 2]};
            const float d[3] = {rays_d[3 * r], rays_d[3 * r + 1], rays_d[3 * r + 2]};
            const float c[3] = {cyl[5 * g], cyl[5 * g + 1], cyl[5 * g + 2]};
            if (cylinder_bounds(o, d, c, near_in ? near_in[r] : near0, far_in ? far_in[r] : far0, &nr[k], &fr[k])) {
                sn += (double)nr[k]; sf += (double)fr[k]; sc += 1.0;
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        sn += __shfl_xor(sn, off, 64);
        sf += __shfl_xor(sf, off, 64);
        sc += __shfl_xor(sc, off, 64);
    }
    if ((threadIdx.x & 63) == 0) { s_part[threadIdx.x >> 6][0] = sn; s_part[threadIdx.x >> 6][1] = sf; s_part[threadIdx.x >> 6][2] = sc; }
    __syncthreads();
    if (threadIdx.x < 3) {
        double t = 0.0;
        for (int w = 0; w < CYL_BLOCK / 64; ++w) t += s_part[w][threadIdx.x];
        s_sum[threadIdx.x] = t;
    }
    __syncthreads();
    const double cnt = s_sum[2];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int r = r_begin + k * CYL_BLOCK + (int)threadIdx.x;
        if (r < r_end) {
            float n = nr[k], f = fr[k];
            if (n != n) {  // ray missed the cylinder: chunk-wide nan-mean (ray_utils.py:330-344)
                n = cnt > 0.0 ? (float)(s_sum[0] / cnt) : (near_in ? near_in[r] : near0);
                f = cnt > 0.0 ? (float)(s_sum[1] / cnt) : (far_in ? far_in[r] : far0);
            }
            near_out[r] = n;
            far_out[r] = f;
        }
    }
}

// ======================================================================================
// per-bone box near / far (fast configs)
// ======================================================================================
__global__ __launch_bounds__(256) void k_box_bounds(const float* __restrict__ rays_o,
                                                    const float* __restrict__ rays_d,
                                                    const float* __restrict__ skts,
                                                    const float* __restrict__ align,
                                                    const float* __restrict__ axis_scale, int R, int G,
                                                    float* __restrict__ near_io, float* __restrict__ far_io) {
    __shared__ float s_align[J * 16];
    __shared__ float s_scale[J * 3];
    for (int i = threadIdx.x; i < J * 16; i += blockDim.x) s_align[i] = align[i];
    for (int i = threadIdx.x; i < J * 3; i += blockDim.x) s_scale[i] = fabsf(axis_scale[i]);
    __syncthreads();
    const int rays_per_pose = R / G;
    // one lane per (ray, bone): 32 lanes per ray (24 active), min / max over the bones by half-wave shuffles -- the same values
    // as the sequential loop (min / max do not depend on the order); a thread per ray walking 24 fp64 slab tests was
    // latency-bound (33 us for 3 072 rays)
    const int j = threadIdx.x & 31;
    const long slots = ((long)gridDim.x * blockDim.x) >> 5;
    for (long r0 = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 5; r0 < ((R + 1) & ~1L); r0 += slots) {   // both halves of a wave iterate together
        const int r = (int)(r0 < R ? r0 : R - 1);
        const int g = min(r / rays_per_pose, G - 1);
        float lo = 100000.0f, hi = -100000.0f;
        bool ok = false;
        if (j < J && r0 < R) {
            const float o[3] = {rays_o[3 * r], rays_o[3 * r + 1], rays_o[3 * r + 2]};
            const float d[3] = {rays_d[3 * r], rays_d[3 * r + 1], rays_d[3 * r + 2]};
            float sk[12];
            const float* src = skts + ((size_t)g * J + j) * 16;
#pragma unroll
            for (int i = 0; i < 12; ++i) sk[i] = src[i];
            float l_, h_;
            ok = bone_box_steps(sk, s_align + 16 * j, s_scale + 3 * j, o, d, &l_, &h_);
            if (ok) { lo = l_; hi = h_; }
        }
        int any = ok ? 1 : 0;
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) {
            lo = fminf(lo, __shfl_xor(lo, off, 64));
            hi = fmaxf(hi, __shfl_xor(hi, off, 64));
            any |= __shfl_xor(any, off, 64);
        }
        if (j == 0 && r0 < R && any) {
            near_io[r] = lo;
            far_io[r] = hi;
        }
    }
}

// ======================================================================================
// coarse samples
// ======================================================================================
__device__ __forceinline__ float coarse_sample(float n_, float f_, int s, int S, const float* __restrict__ t_rand, long m) {
    float v = coarse_z(n_, f_, s, S);
    if (t_rand) {  // stratified jitter (ray_utils.py:233-248)
        const float zp = s > 0 ? coarse_z(n_, f_, s - 1, S) : v;
        const float zn = s + 1 < S ? coarse_z(n_, f_, s + 1, S) : v;
        const float lower = s > 0 ? mul_rn(0.5f, add_rn(v, zp)) : v;
        const float upper = s + 1 < S ? mul_rn(0.5f, add_rn(zn, v)) : v;
        v = add_rn(lower, mul_rn(sub_rn(upper, lower), t_rand[m]));
    }
    return v;
}

// S a multiple of 4: four depths of a ray per thread, one 16-byte store (the frame's 50 MB of depths are a pure write stream: 4-byte
// stores reached 2.9 TB/s of the 6 a fill reaches)
__global__ __launch_bounds__(256) void k_coarse_samples4(const float* __restrict__ nr, const float* __restrict__ fr,
                                                         int R, int S, const float* __restrict__ t_rand,
                                                         float4* __restrict__ z) {
    const unsigned S4 = (unsigned)S / 4u;
    const unsigned M4 = (unsigned)R * S4;
    for (unsigned q = blockIdx.x * blockDim.x + threadIdx.x; q < M4; q += gridDim.x * blockDim.x) {
        const unsigned r = q / S4;
        const int s0 = 4 * (int)(q - r * S4);
        const float n_ = nr[r], f_ = fr[r];
        const long m = 4L * q;
        float4 v;
        v.x = coarse_sample(n_, f_, s0, S, t_rand, m);
        v.y = coarse_sample(n_, f_, s0 + 1, S, t_rand, m + 1);
        v.z = coarse_sample(n_, f_, s0 + 2, S, t_rand, m + 2);
        v.w = coarse_sample(n_, f_, s0 + 3, S, t_rand, m + 3);
        z[q] = v;
    }
}

__global__ __launch_bounds__(256) void k_coarse_samples(const float* __restrict__ nr, const float* __restrict__ fr,
                                                        int R, int S, const float* __restrict__ t_rand,
                                                        float* __restrict__ z) {
    const long M = (long)R * S;
    const bool small = M <= 0x7fffffffL;        // 32-bit index arithmetic (a 64-bit division per sample was most of this kernel)
    for (long m = (long)blockIdx.x * blockDim.x + threadIdx.x; m < M; m += (long)gridDim.x * blockDim.x) {
        int r, s;
        if (small) {
            r = (int)((unsigned)m / (unsigned)S);
            s = (int)((unsigned)m - (unsigned)r * (unsigned)S);
        } else {
            r = (int)(m / S);
            s = (int)(m % S);
        }
        z[m] = coarse_sample(nr[r], fr[r], s, S, t_rand, m);
    }
}

// sample point m: either o + d*z (two roundings, core/raycasters.py:463) or read from pts
__device__ __forceinline__ void load_point(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                           const float* __restrict__ z, const float* __restrict__ pts, long m, int S,
                                           float* p) {
    if (pts != nullptr) {
        p[0] = pts[3 * m]; p[1] = pts[3 * m + 1]; p[2] = pts[3 * m + 2];
    } else {
        const int r = (int)(m / S);
        const float o[3] = {rays_o[3 * r], rays_o[3 * r + 1], rays_o[3 * r + 2]};
        const float d[3] = {rays_d[3 * r], rays_d[3 * r + 1], rays_d[3 * r + 2]};
        sample_point(o, d, z[m], p);
    }
}

// ======================================================================================
// K1a: transform + cull
// ======================================================================================
typedef float v2f __attribute__((ext_vector_type(2)));

// ((m0*x + m1*y) + m2*z) + m3 for two points at once; every packed op rounds each half like the
// scalar __fmul_rn / __fadd_rn chain of affine_unfused() (contraction disabled).
// The matrix entries are BROADCAST over the pair.  Left to the compiler that is v_pk_mul_f32 x, m op_sel:[0,1] for the odd entries --
// the low half from SRC1's high dword: the one operand selection that is wrong on gfx950 beside MFMA wavefronts (common.hpp,
// DANBO_NO_PK_F32; without packed instructions this kernel takes 102 instead of 55 us).  Written out with the matrix pair as SRC0, whose
// high-dword selection is exact (tools/probe/cview_probe.hip: 0 of 2.7e10), the products and sums are the same IEEE operations.
template <int HI>
__device__ __forceinline__ v2f pk_mul_bcast(v2f m, v2f x) {     // m[HI] * x
    v2f d;
    if (HI) asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(d) : "v"(m), "v"(x));
    else asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(d) : "v"(m), "v"(x));
    return d;
}
__device__ __forceinline__ v2f pk_add(v2f a, v2f b) {
    v2f d;
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ v2f pk_add_bcast_hi(v2f m, v2f a) {  // m[1] + a
    v2f d;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(d) : "v"(m), "v"(a));
    return d;
}
__device__ __forceinline__ void affine_pk(const float* M, v2f x, v2f y, v2f z, v2f* q) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float4 r = *reinterpret_cast<const float4*>(M + 4 * k);
        const v2f m01 = {r.x, r.y}, m23 = {r.z, r.w};
        v2f s = pk_add(pk_mul_bcast<0>(m01, x), pk_mul_bcast<1>(m01, y));
        s = pk_add(s, pk_mul_bcast<0>(m23, z));
        q[k] = pk_add_bcast_hi(m23, s);
    }
}

// Conservative ray-level rejection shared by k_ray_bone_mask and the in-kernel prefilter of k_bone_cull: true when the
// segment {o + t d : zl - pad <= t <= zh + pad} provably misses the slightly inflated box of the bone (margins far above the
// fp32 round-off of the transform chain).  NaN anywhere -> not provably missed -> false.
__device__ __forceinline__ bool segment_misses_bone(const float* sk, const float* al, const float* sc /*|axis scale|, 3*/,
                                                    const float* o, const float* d, float zl, float zh) {
    float ol[3], dl[3], t[3];
    bone_local(sk, al, o, ol);
    for (int k = 0; k < 3; ++k) t[k] = sk[4 * k] * d[0] + sk[4 * k + 1] * d[1] + sk[4 * k + 2] * d[2];
    for (int k = 0; k < 3; ++k) dl[k] = al[4 * k] * t[0] + al[4 * k + 1] * t[1] + al[4 * k + 2] * t[2];
    const float pad = 1e-4f * fmaxf(fabsf(zl), fabsf(zh)) + 1e-5f;
    float tmin = zl - pad, tmax = zh + pad;
    bool miss = false;
    for (int k = 0; k < 3; ++k) {
        const float s = sc[k] * 1.001f + 1e-4f;
        if (fabsf(dl[k]) < 1e-12f) {
            miss = miss || fabsf(ol[k]) > s;
        } else {
            const float inv = 1.0f / dl[k];
            const float t1 = (-s - ol[k]) * inv, t2 = (s - ol[k]) * inv;
            tmin = fmaxf(tmin, fminf(t1, t2));
            tmax = fminf(tmax, fmaxf(t1, t2));
        }
    }
    return miss || tmin > tmax;
}

// ray_mask[r] bit j = 0: no point o + t d of ray r with t_lo[r] <= t <= t_hi[r] (+ the pad above) can lie inside the volume of
// bone j.  Once per frame -- both sampling passes of a ray stay inside its [near, far].  A thread per ray walks the 24 bones out
// of LDS (the poses of the workgroup's 256 rays are staged: POSES_IN_LDS; otherwise the matrices come from global memory): no
// memory round trip inside the loop.  (A lane per (ray, bone) as k_box_bounds, matrices from global: 48 us for 262 144 rays, all
// of it load latency; this: see DESIGN.md section 3.)
// No reference counterpart (the reference tests every sample against every bone, gnn_backbone.py:787-828): the exact per-sample
// test of k_bone_cull is unchanged, this only tells it which bones (and which whole workgroups) cannot matter.
constexpr float RAY_FLAT_VMAX = DANBO_RAY_FLAT_VMAX;

template <bool POSES_IN_LDS>
__global__ __launch_bounds__(256) void k_ray_bone_mask(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                       const float* __restrict__ t_lo, const float* __restrict__ t_hi,
                                                       const float* __restrict__ skts, const float* __restrict__ align,
                                                       const float* __restrict__ axis_scale, int R, int G,
                                                       uint32_t* __restrict__ ray_mask, uint32_t* __restrict__ ray_flat) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_align = smem;                 // [24][16]
    float* s_scale = smem + J * 16;        // [24][4]
    float* s_skt = smem + J * 16 + J * 4;  // [poses of this workgroup][24][16]
    const int rays_per_pose = R / G;
    const int r_a = blockIdx.x * 256, r_b = min(r_a + 255, R - 1);
    const int g0 = min(r_a / rays_per_pose, G - 1), g1 = min(r_b / rays_per_pose, G - 1);
    // this thread's ray first: its loads fly while the matrices are staged
    const int r = min(r_a + (int)threadIdx.x, R - 1);
    const float o[3] = {rays_o[3 * r], rays_o[3 * r + 1], rays_o[3 * r + 2]};
    const float d[3] = {rays_d[3 * r], rays_d[3 * r + 1], rays_d[3 * r + 2]};
    const float zl = t_lo[r], zh = t_hi[r];
    for (int i = threadIdx.x; i < J * 16; i += 256) s_align[i] = align[i];
    for (int i = threadIdx.x; i < J * 4; i += 256) s_scale[i] = (i & 3) < 3 ? fabsf(axis_scale[(i >> 2) * 3 + (i & 3)]) : 0.f;
    if (POSES_IN_LDS)
        for (int i = threadIdx.x; i < (g1 - g0 + 1) * J * 16; i += 256) s_skt[i] = skts[(size_t)g0 * J * 16 + i];
    __syncthreads();
    const int g = min(r / rays_per_pose, G - 1);
    uint32_t bits = 0u;
    if (POSES_IN_LDS) {
        const float* sk = s_skt + (size_t)(g - g0) * J * 16;
#pragma unroll 2
        for (int j = 0; j < J; ++j)
            bits |= segment_misses_bone(sk + 16 * j, s_align + 16 * j, s_scale + 4 * j, o, d, zl, zh) ? 0u : 1u << j;
    } else {
        for (int j = 0; j < J; ++j) {
            float sk[12];
            const float* src = skts + ((size_t)g * J + j) * 16;
#pragma unroll
            for (int i = 0; i < 12; ++i) sk[i] = src[i];
            bits |= segment_misses_bone(sk, s_align + 16 * j, s_scale + 4 * j, o, d, zl, zh) ? 0u : 1u << j;
        }
    }
    if (r_a + (int)threadIdx.x < R) {
        ray_mask[r] = bits;
        if (ray_flat != nullptr) {
            // candidate for a ray of constants (k_flat_rays): no volume anywhere along [zl, zh], every interval length any set of
            // depths inside [zl, zh] can produce -- |gap| * |d|, 1e10 * |d| for the last sample -- is finite, and every input the
            // view layer can see of this ray (its direction, raw or turned by the root bone's matrix; sines and cosines are
            // <= 1 anyway) is at most RAY_FLAT_VMAX in magnitude: what the caller's bound on the empty-space colour assumes
            const float dn = norm3_torch(d[0], d[1], d[2]);
            const float span = mul_rn(add_rn(sub_rn(zh, zl), mul_rn(1e-3f, fmaxf(fabsf(zl), fabsf(zh)))), dn);
            const float tail = mul_rn(1e10f, dn);
            const float* m0 = POSES_IN_LDS ? s_skt + (size_t)(g - g0) * J * 16 : skts + (size_t)g * J * 16;     // the root bone
            float msum = 0.f;
            for (int a = 0; a < 3; ++a)
                for (int k = 0; k < 3; ++k) msum = add_rn(msum, fabsf(m0[4 * a + k]));
            const float vin = mul_rn(dn, fmaxf(1.0f, msum));
            const bool fin = zl <= zh && sub_rn(span, span) == 0.f && sub_rn(tail, tail) == 0.f && vin <= RAY_FLAT_VMAX;
            ray_flat[r] = (bits == 0u && fin) ? 1u : 0u;
        }
    }
}

constexpr int CULL_BLOCK = 256;
constexpr int CULL_SPT = 4;  // samples per thread -> 1024 consecutive samples per workgroup
constexpr int CULL_MAX_RAYS = 130;  // rays a workgroup may span (S >= 8) for the ray-level bone rejection

__global__ __launch_bounds__(CULL_BLOCK) void k_bone_cull(const float* __restrict__ rays_o,
                                                          const float* __restrict__ rays_d,
                                                          const float* __restrict__ z,
                                                          const float* __restrict__ pts, int R, int S, int G,
                                                          const float* __restrict__ skts,
                                                          const float* __restrict__ align,
                                                          const float* __restrict__ axis_scale, int np_lds,
                                                          const uint32_t* __restrict__ ray_mask,
                                                          const float* __restrict__ t_lo, const float* __restrict__ t_hi,
                                                          uint32_t* __restrict__ ray_flat,
                                                          uint32_t* __restrict__ valid_bits,
                                                          int32_t* __restrict__ list, int32_t* __restrict__ count) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_align = smem;                 // [24][16]
    float* s_scale = smem + J * 16;        // [24][4]
    float* s_skt = smem + J * 16 + J * 4;  // [np_lds][24][16]
    __shared__ float s_zlo[CULL_MAX_RAYS], s_zhi[CULL_MAX_RAYS];
    __shared__ uint32_t s_mask[CULL_MAX_RAYS];

    const long M = (long)R * S;
    const long spp = (long)(R / G) * S;  // samples per pose
    const long base = (long)blockIdx.x * (CULL_BLOCK * CULL_SPT);
    const long last = min(base + CULL_BLOCK * CULL_SPT, M) - 1;
    const int g0 = (int)min(base / spp, (long)G - 1);
    const int g1 = (int)min(last / spp, (long)G - 1);

    const int lane = threadIdx.x & 63;
    // 32-bit bookkeeping of the workgroup's 1024 samples: sample k * 256 + tid of the workgroup is ray r_first + li / S with
    // li = off0 + k * 256 + tid < S + 1024 (one 64-bit division per workgroup instead of a dozen per thread -- the emulated 64-bit
    // divides of m / S and m / spp were most of this kernel's instructions), pose = ray / rays_per_pose
    const int r_first = (int)(base / S), r_last = (int)(last / S);
    const int off0 = (int)(base - (long)r_first * S);
    const int rays_per_pose = R / G;
    const float inv_S = 1.0f / (float)S;
    auto ray_of = [&](int li) {          // r_first + li / S, exact for li < 2^22
        int q = (int)((float)li * inv_S);
        q -= (q * S > li) ? 1 : 0;
        q += ((q + 1) * S <= li) ? 1 : 0;
        return q;
    };
    auto pose_of = [&](int r) { return G == 1 ? 0 : min(r / rays_per_pose, G - 1); };
    // ---- ray-level rejection (z mode): a bone whose slightly inflated box the ray's sampled segment misses
    // cannot contain any of the ray's samples, so whole wavefronts skip it.  Conservative (margins far above
    // fp32 round-off of the transform chain); the per-sample test below is unchanged, so the mask stays exact.
    const int nrays = r_last - r_first + 1;
    const bool prefilter = pts == nullptr && ray_mask == nullptr && nrays <= CULL_MAX_RAYS;
    // this thread's depths (z mode): sample k * 256 + tid of the workgroup -- read once, used by the ray extents and by the points
    float zreg[CULL_SPT];
    if (pts == nullptr) {
#pragma unroll
        for (int k = 0; k < CULL_SPT; ++k) zreg[k] = z[min(base + k * CULL_BLOCK + threadIdx.x, M - 1)];
    }
    // ---- with a per-ray bone mask (k_ray_bone_mask, once per frame): candidate bones of every sample without touching a
    // matrix.  A depth outside the interval the mask was made for falls back to all bones (the mask stays exact whatever the
    // caller passes); a workgroup none of whose samples has a candidate -- most of a frame: the body covers a tenth of the
    // image -- stores its zeros and leaves before staging anything.
    uint32_t pre[CULL_SPT];
    if (ray_mask != nullptr) {
        bool any = false;
#pragma unroll
        for (int k = 0; k < CULL_SPT; ++k) {
            pre[k] = 0u;
            if (base + k * CULL_BLOCK + threadIdx.x < M) {
                const int r = r_first + ray_of(off0 + k * CULL_BLOCK + threadIdx.x);
                const float lo = t_lo[r], hi = t_hi[r];
                const float slack = 5e-5f * fmaxf(fabsf(lo), fabsf(hi));   // half of segment_misses_bone's pad
                const bool inside = zreg[k] >= lo - slack && zreg[k] <= hi + slack;
                pre[k] = inside ? ray_mask[r] : (1u << J) - 1u;
                if (!inside && ray_flat != nullptr) ray_flat[r] = 0u;     // not a ray of constants after all (every writer stores 0)
            }
            any = any || pre[k] != 0u;
        }
        if (!__syncthreads_or(any ? 1 : 0)) {
#pragma unroll
            for (int k = 0; k < CULL_SPT; ++k)
                if (base + k * CULL_BLOCK + threadIdx.x < M) valid_bits[base + k * CULL_BLOCK + threadIdx.x] = 0u;
            return;
        }
    }
    for (int i = threadIdx.x; i < J * 16; i += CULL_BLOCK) s_align[i] = align[i];
    for (int i = threadIdx.x; i < J * 4; i += CULL_BLOCK) s_scale[i] = (i & 3) < 3 ? fabsf(axis_scale[(i >> 2) * 3 + (i & 3)]) : 0.f;
    const int npose = g1 - g0 + 1;  // <= np_lds by construction of the launch
    for (int i = threadIdx.x; i < npose * J * 16; i += CULL_BLOCK) s_skt[i] = skts[(size_t)g0 * J * 16 + i];
    __syncthreads();

    if (prefilter) {
        // extent [min z, max z] of every ray's samples in this workgroup's window -- the samples of a ray are either sorted
        // (coarse / deterministic importance) or not (random draws): LDS atomics on an order-preserving integer image of the
        // floats, all 256 threads (one thread per ray walking its S depths one after the other was a chain of S loads)
        int* s_lo = reinterpret_cast<int*>(s_zlo);
        int* s_hi = reinterpret_cast<int*>(s_zhi);
        for (int i = threadIdx.x; i < nrays; i += CULL_BLOCK) { s_lo[i] = INT_MAX; s_hi[i] = INT_MIN; s_mask[i] = 0u; }
        __syncthreads();
        auto ordered = [](float f) { const int b = __builtin_bit_cast(int, f); return b >= 0 ? b : (int)(0x80000000u - (unsigned)b); };
        auto unordered = [](int o) { return __builtin_bit_cast(float, o >= 0 ? o : (int)(0x80000000u - (unsigned)o)); };
#pragma unroll
        for (int k = 0; k < CULL_SPT; ++k) {
            if (base + k * CULL_BLOCK + threadIdx.x < M) {
                const int i = ray_of(off0 + k * CULL_BLOCK + threadIdx.x);
                const int oz = ordered(zreg[k]);
                atomicMin(&s_lo[i], oz);
                atomicMax(&s_hi[i], oz);
            }
        }
        __syncthreads();
        // A window that starts or ends inside a ray sees only part of that ray's samples: its extent here covers exactly the
        // samples THIS workgroup tests, which is all the rejection needs.
        for (int i = threadIdx.x; i < nrays; i += CULL_BLOCK) {
            const float lo = unordered(s_lo[i]), hi = unordered(s_hi[i]);
            s_zlo[i] = lo; s_zhi[i] = hi;
        }
        __syncthreads();
        for (int q = threadIdx.x; q < nrays * J; q += CULL_BLOCK) {
            const int i = q / J, j = q % J, r = r_first + i;
            const int g = pose_of(r);
            const float* sk = s_skt + (g - g0) * J * 16 + 16 * j;
            const float* al = s_align + 16 * j;
            const float o[3] = {rays_o[3 * r], rays_o[3 * r + 1], rays_o[3 * r + 2]};
            const float d[3] = {rays_d[3 * r], rays_d[3 * r + 1], rays_d[3 * r + 2]};
            const bool miss = segment_misses_bone(sk, al, s_scale + 4 * j, o, d, s_zlo[i], s_zhi[i]);
            if (!miss) atomicOr(&s_mask[i], 1u << j);
        }
        __syncthreads();
    }
    // Two samples per lane and per instruction: the unfused mul/add chain runs on packed fp32
    // (v_pk_mul_f32 / v_pk_add_f32: two IEEE-rounded results per lane-op, same values as the scalar
    // chain), the per-bone matrices are LDS broadcasts splatted over the pair.
    uint32_t bits4[CULL_SPT];
#pragma unroll
    for (int pr = 0; pr < CULL_SPT / 2; ++pr) {
        const long ma = base + (2 * pr) * CULL_BLOCK + threadIdx.x;
        const long mb = ma + CULL_BLOCK;
        const long mac = min(ma, M - 1), mbc = min(mb, M - 1);
        // (clamped tail samples repeat the last sample: local index of the clamp)
        const int lia = (int)(mac - base) + off0, lib = (int)(mbc - base) + off0;
        const int ra = r_first + ray_of(lia), rb = r_first + ray_of(lib);
        float pa[3], pb[3];
        if (pts != nullptr) {
            pa[0] = pts[3 * mac]; pa[1] = pts[3 * mac + 1]; pa[2] = pts[3 * mac + 2];
            pb[0] = pts[3 * mbc]; pb[1] = pts[3 * mbc + 1]; pb[2] = pts[3 * mbc + 2];
        } else {
            const float oa[3] = {rays_o[3 * ra], rays_o[3 * ra + 1], rays_o[3 * ra + 2]};
            const float da[3] = {rays_d[3 * ra], rays_d[3 * ra + 1], rays_d[3 * ra + 2]};
            const float ob[3] = {rays_o[3 * rb], rays_o[3 * rb + 1], rays_o[3 * rb + 2]};
            const float db[3] = {rays_d[3 * rb], rays_d[3 * rb + 1], rays_d[3 * rb + 2]};
            // (a clamped tail sample re-reads the last depth, as z[mac] did)
            sample_point(oa, da, ma < M ? zreg[2 * pr] : z[M - 1], pa);
            sample_point(ob, db, mb < M ? zreg[2 * pr + 1] : z[M - 1], pb);
        }
        const int ga = pose_of(ra), gb = pose_of(rb);
        uint32_t ba = 0, bb = 0;
        // bones any lane of this wavefront may be inside (wave-uniform)
        uint32_t need = (1u << J) - 1u;
        if (ray_mask != nullptr) {
            need = wave_or(pre[2 * pr] | pre[2 * pr + 1]);
        } else if (prefilter) {
            uint32_t mine = s_mask[ra - r_first] | s_mask[rb - r_first];
            need = wave_or(mine);
        }
        if (__builtin_amdgcn_readfirstlane((int)__all(ga == gb))) {
            const v2f px = {pa[0], pb[0]}, py = {pa[1], pb[1]}, pz = {pa[2], pb[2]};
            const float* sk = s_skt + (ga - g0) * J * 16;
            while (need) {
                const int j = __builtin_ctz(need);
                need &= need - 1u;
                v2f l[3], t[3];
                affine_pk(sk + 16 * j, px, py, pz, l);
                affine_pk(s_align + 16 * j, l[0], l[1], l[2], t);
                const float* sc = s_scale + 4 * j;
                const bool ina = !(fabsf(t[0].x) > sc[0] || fabsf(t[1].x) > sc[1] || fabsf(t[2].x) > sc[2]);
                const bool inb = !(fabsf(t[0].y) > sc[0] || fabsf(t[1].y) > sc[1] || fabsf(t[2].y) > sc[2]);
                ba |= (ina ? 1u : 0u) << j;
                bb |= (inb ? 1u : 0u) << j;
            }
        } else {  // some pair of the wavefront straddles two poses (chunk boundaries only)
            const float* ska = s_skt + (ga - g0) * J * 16;
            const float* skb = s_skt + (gb - g0) * J * 16;
            for (int j = 0; j < J; ++j) {
                float pt[3];
                bone_local(ska + 16 * j, s_align + 16 * j, pa, pt);
                ba |= (in_volume(pt, s_scale + 4 * j) ? 1u : 0u) << j;
                bone_local(skb + 16 * j, s_align + 16 * j, pb, pt);
                bb |= (in_volume(pt, s_scale + 4 * j) ? 1u : 0u) << j;
            }
        }
        bits4[2 * pr] = ma < M ? ba : 0u;
        bits4[2 * pr + 1] = mb < M ? bb : 0u;
        if (ma < M) valid_bits[ma] = ba;
        if (mb < M) valid_bits[mb] = bb;
    }
    if (list != nullptr) {
        // one atomic per WORKGROUP: the 1024 consecutive samples of this workgroup stay in order in the
        // list (good locality for the per-wavefront bone skipping of K2), order across workgroups free
        __shared__ int s_cnt[CULL_SPT * (CULL_BLOCK / 64)];
        __shared__ int s_base;
        const int wave = threadIdx.x >> 6;
        unsigned long long ball[CULL_SPT];
#pragma unroll
        for (int it = 0; it < CULL_SPT; ++it) {
            ball[it] = __ballot(bits4[it] != 0);
            if (lane == 0) s_cnt[it * (CULL_BLOCK / 64) + wave] = __popcll(ball[it]);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            int run = 0;
            for (int i = 0; i < CULL_SPT * (CULL_BLOCK / 64); ++i) { const int c = s_cnt[i]; s_cnt[i] = run; run += c; }
            s_base = run > 0 ? atomicAdd(count, run) : 0;
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < CULL_SPT; ++it) {
            if (bits4[it] != 0) {
                const long m = base + it * CULL_BLOCK + threadIdx.x;
                list[s_base + s_cnt[it * (CULL_BLOCK / 64) + wave] + __popcll(ball[it] & ((1ull << lane) - 1ull))] = (int32_t)m;
            }
        }
    }
}

// ======================================================================================
// K1b: factorised gather -> part_feat [n,24,15]
// ======================================================================================
constexpr int GATHER_TS = 16;               // samples per tile
constexpr int GATHER_BLOCK = GATHER_TS * J;  // 384 threads: one (sample, bone) item each

__global__ __launch_bounds__(GATHER_BLOCK) void k_bone_gather(const float* __restrict__ rays_o,
                                                              const float* __restrict__ rays_d,
                                                              const float* __restrict__ z,
                                                              const float* __restrict__ pts, int R, int S, int G,
                                                              const float* __restrict__ skts,
                                                              const float* __restrict__ align,
                                                              const float* __restrict__ axis_scale,
                                                              const float* __restrict__ volumes,
                                                              const int32_t* __restrict__ list,
                                                              const int32_t* __restrict__ count, int n_cap,
                                                              float* __restrict__ part_feat) {
    __shared__ __attribute__((aligned(16))) float s_vol[J * VOL];          // 23040 B
    // (round 6 measured the output tile double-buffered -- tile i stored while tile i + 1 is computed, one barrier per tile instead
    // of two: 72 KB of LDS leave two workgroups per CU instead of three and the launch drops from 3.6 to 2.7 TB/s; not kept)
    __shared__ __attribute__((aligned(16))) float s_out[GATHER_TS * J * FEAT];  // 23040 B
    __shared__ __attribute__((aligned(16))) float s_skt[J * 16];
    __shared__ __attribute__((aligned(16))) float s_align[J * 16];
    __shared__ float s_scale[J * 4];

    const int n = resolve_count(count, n_cap);
    const int ntiles = (n + GATHER_TS - 1) / GATHER_TS;
    const long spp = (long)(R / G) * S;
    const int tid = threadIdx.x;
    const int sl = tid / J, j = tid % J;

    for (int i = tid; i < J * 16; i += GATHER_BLOCK) s_align[i] = align[i];
    for (int i = tid; i < J * 4; i += GATHER_BLOCK) s_scale[i] = (i & 3) < 3 ? fabsf(axis_scale[(i >> 2) * 3 + (i & 3)]) : 1.f;
    int g_lds = -1;

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int row0 = tile * GATHER_TS;
        const int first = list ? list[row0] : row0;
        const int g_tile = (int)min((long)first / spp, (long)G - 1);
        __syncthreads();  // previous tile's s_out fully stored, s_vol no longer read
        if (g_tile != g_lds) {
            const float4* src = reinterpret_cast<const float4*>(volumes + (size_t)g_tile * J * VOL);
            float4* dst = reinterpret_cast<float4*>(s_vol);
            for (int i = tid; i < J * VOL / 4; i += GATHER_BLOCK) dst[i] = src[i];
            for (int i = tid; i < J * 16; i += GATHER_BLOCK) s_skt[i] = skts[(size_t)g_tile * J * 16 + i];
            g_lds = g_tile;
        }
        __syncthreads();
        const int row = row0 + sl;
        float* o_ = s_out + (sl * J + j) * FEAT;
        if (row < n) {
            const long m = list ? list[row] : row;
            const int g = (int)min(m / spp, (long)G - 1);
            float p[3], pt[3], f[FEAT];
            load_point(rays_o, rays_d, z, pts, m, S, p);
            if (g == g_lds) {
                bone_local(s_skt + 16 * j, s_align + 16 * j, p, pt);
                gather_bone_features(s_vol + j * VOL, pt, s_scale + 4 * j, f);
            } else {  // tile straddles two poses (multi-pose chunks only): read through L1/L2
                float sk[12];
                const float* src = skts + ((size_t)g * J + j) * 16;
#pragma unroll
                for (int i = 0; i < 12; ++i) sk[i] = src[i];
                bone_local(sk, s_align + 16 * j, p, pt);
                gather_bone_features(volumes + ((size_t)g * J + j) * VOL, pt, s_scale + 4 * j, f);
            }
#pragma unroll
            for (int k = 0; k < FEAT; ++k) o_[k] = f[k];
        } else {
#pragma unroll
            for (int k = 0; k < FEAT; ++k) o_[k] = 0.f;
        }
        __syncthreads();
        // coalesced 16-B stores of the 16 x 1440 B tile
        const int rows_here = min(GATHER_TS, n - row0);
        const int nvec = rows_here * (J * FEAT / 4);
        typedef float f32x4_t __attribute__((ext_vector_type(4)));
        f32x4_t* dst = reinterpret_cast<f32x4_t*>(part_feat + (size_t)row0 * J * FEAT);
        const f32x4_t* src = reinterpret_cast<const f32x4_t*>(s_out);
        // write-once stream: non-temporal, it is 1 440 B per sample and must not evict the volumes / transforms
        for (int i = tid; i < nvec; i += GATHER_BLOCK) __builtin_nontemporal_store(src[i], dst + i);
    }
}

// ======================================================================================
// raw fill / merge
// ======================================================================================
__global__ __launch_bounds__(256) void k_fill_raw(const float4* __restrict__ raw_empty, int R, int S,
                                                  float4* __restrict__ raw) {
    const long M = (long)R * S;
    for (long m = (long)blockIdx.x * blockDim.x + threadIdx.x; m < M; m += (long)gridDim.x * blockDim.x)
        raw[m] = raw_empty[m / S];
}

__global__ __launch_bounds__(256) void k_merge_samples(const float* __restrict__ a, const float* __restrict__ b,
                                                       const int32_t* __restrict__ idx, int R, int S, int Sf, int C,
                                                       float* __restrict__ out) {
    const int St = S + Sf;
    const long N = (long)R * St;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (long)gridDim.x * blockDim.x) {
        const int r = (int)(i / St);
        const int src = min(max(idx[i], 0), St - 1);
        const float* s = src < S ? a + ((size_t)r * S + src) * C : b + ((size_t)r * Sf + (src - S)) * C;
        float* o = out + (size_t)i * C;
        if (C == 4) {
            *reinterpret_cast<float4*>(o) = *reinterpret_cast<const float4*>(s);
        } else {
            for (int c = 0; c < C; ++c) o[c] = s[c];
        }
    }
}

// ======================================================================================
// K4: alpha compositing, one wavefront per ray (lane = sample), scans on the DPP paths
// ======================================================================================
struct CompositeState {
    float carry, sr, sg, sb, sd, sa;
};

// Item g of a launch -> ray (g * scatter) mod R, scatter coprime to R (ray_scatter below): a bijection that spreads the rays of
// an image region over all wavefronts.  Wavefront w walks items w, w + nwaves, ...; with rays in image order and nwaves a
// multiple of the image width those are the pixels of ONE column -- the wavefronts of the columns that cross the body did all
// the work of the frame while the others found nothing but rays of constants (k_composite_importance: 217 us for 37 % of the
// rays, 265 us for all of them).
__device__ __forceinline__ int scattered_ray(long g, unsigned scatter, int R) {
    return (int)((unsigned long long)g * scatter % (unsigned)R);
}

// one 64-sample chunk of a ray: rw = raw of this lane's sample, zs its depth, gap = z[s+1] - z[s] (1e10 for the
// last sample of the ray), nz = optional density noise.  Returns the sample's weight; al = its alpha.
__device__ __forceinline__ float composite_chunk(CompositeState& st, const float4 rw, float zs, float gap, float dn,
                                                 float B, bool has_noise, float nz, bool act, int lane, float& al) {
    const float dist = mul_rn(gap, dn);
    float sig = div_rn(rw.w, B);
    if (has_noise) sig = add_rn(sig, nz);
    sig = fmaxf(sig, 0.f);
    al = sub_rn(1.0f, expf(-mul_rn(sig, dist)));
    if (!act) al = 0.f;
    const float t = act ? add_rn(sub_rn(1.0f, al), 1e-10f) : 1.0f;
    const float p = wave_scan_mul(t);  // inclusive product
    float excl = __shfl_up(p, 1, 64);
    if (lane == 0) excl = 1.0f;
    const float T = mul_rn(st.carry, excl);
    st.carry = mul_rn(st.carry, __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, p), 63)));
    const float w = mul_rn(al, T);
    const float cr = sub_rn(mul_rn(sigmoidf_(rw.x), 1.002f), 0.001f);
    const float cg = sub_rn(mul_rn(sigmoidf_(rw.y), 1.002f), 0.001f);
    const float cb = sub_rn(mul_rn(sigmoidf_(rw.z), 1.002f), 0.001f);
    st.sr += wave_total(w * cr);
    st.sg += wave_total(w * cg);
    st.sb += wave_total(w * cb);
    st.sd += wave_total(w * zs);
    st.sa += wave_total(w);
    return w;
}

__device__ __forceinline__ void composite_finish(const CompositeState& st, int r, float* __restrict__ rgb_map,
                                                 float* __restrict__ disp, float* __restrict__ acc_out) {
    rgb_map[3 * r] = st.sr;
    rgb_map[3 * r + 1] = st.sg;
    rgb_map[3 * r + 2] = st.sb;
    float dsp = div_rn(1.0f, fmaxf(1e-10f, div_rn(st.sd, add_rn(st.sa, 1e-10f))));
    if (fabsf(st.sa) <= 1e-8f) dsp = 0.f;
    disp[r] = dsp;
    acc_out[r] = fminf(st.sa, 1.0f);
}

__device__ __forceinline__ float ray_norm(const float* __restrict__ rays_d, int r) {
    const float dx = rays_d[3 * r], dy = rays_d[3 * r + 1], dz_ = rays_d[3 * r + 2];
    return norm3_torch(dx, dy, dz_);
}

// raw_empty / bits (optional, together): samples whose in-volume word is 0 were never written by K3 and take the ray's
// empty-space raw (the convention of the fused composites); ray_list / ray_count (optional): only the listed rays
__global__ __launch_bounds__(256) void k_composite(const float4* __restrict__ raw, const float4* __restrict__ raw_empty,
                                                   const uint32_t* __restrict__ bits, const float* __restrict__ z,
                                                   const float* __restrict__ rays_d, int R, int S, float B,
                                                   const float* __restrict__ noise, float* __restrict__ rgb_map,
                                                   float* __restrict__ disp, float* __restrict__ acc_out,
                                                   float* __restrict__ weights, float* __restrict__ alpha_out,
                                                   const int32_t* __restrict__ ray_list, const int32_t* __restrict__ ray_count) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    const int n = ray_list ? min(max(*ray_count, 0), R) : R;        // (a list: k_flat_rays' -- only the listed rays are composited)
    for (int i = wave; i < n; i += nwaves) {
        const int r = ray_list ? min(max(ray_list[i], 0), R - 1) : i;
        const float dn = ray_norm(rays_d, r);
        const float4 re = bits ? raw_empty[r] : float4{0.f, 0.f, 0.f, 0.f};
        CompositeState st = {1.0f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int c0 = 0; c0 < S; c0 += 64) {
            const int s = c0 + lane;
            const bool act = s < S;
            const size_t m = (size_t)r * S + (act ? s : S - 1);
            const float zs = z[m];
            const float gap = (s + 1 < S) ? sub_rn(z[m + 1], zs) : 1e10f;
            float4 rw = re;
            if (bits == nullptr || bits[m] != 0u) rw = raw[m];
            float al;
            const float w = composite_chunk(st, rw, zs, gap, dn, B, noise != nullptr, noise ? noise[m] : 0.f, act, lane, al);
            if (act) {
                if (weights) weights[m] = w;
                if (alpha_out) alpha_out[m] = al;
            }
        }
        if (lane == 0) composite_finish(st, r, rgb_map, disp, acc_out);
    }
}

// The last composite of the two-pass render without materialising the merged raw tensor: sample i of the sorted
// order is fetched from the coarse (src < S) or the importance (src >= S) pass through sorted_idx
// (merge_samples, core/raycasters.py:745-761, folded into raw2outputs).  bits_* / raw_empty (optional): samples
// whose in-volume word is 0 were never written by K3 and take the ray's empty-space raw instead.
// (amdgpu_waves_per_eu(8, 8): the compiler's 103 - 106 SGPRs allowed seven wavefronts per SIMD; a ray is a chain of dependent memory
// round trips that only other wavefronts cover -- with 78 SGPRs and eight: 201 -> 164 us over a whole frame, 303 -> 277 for
// k_composite_importance, the frame -0.4 %)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_composite_merged(const float4* __restrict__ raw_a, const float4* __restrict__ raw_b,
                                                          const float4* __restrict__ raw_empty,
                                                          const uint32_t* __restrict__ bits_a,
                                                          const uint32_t* __restrict__ bits_b,
                                                          const int32_t* __restrict__ sorted_idx,
                                                          const float* __restrict__ z, const float* __restrict__ rays_d,
                                                          int R, int S, int Sf, float B, const float* __restrict__ noise,
                                                          float* __restrict__ rgb_map, float* __restrict__ disp,
                                                          float* __restrict__ acc_out, float* __restrict__ weights,
                                                          float* __restrict__ alpha_out, float4* __restrict__ raw_sorted,
                                                          const int32_t* __restrict__ ray_list,
                                                          const int32_t* __restrict__ ray_count, unsigned scatter) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    const int St = S + Sf;
    if (St <= 64) {
        // one chunk per ray, software-pipelined over the wavefront's rays: a ray is the chain sorted_idx -> in-volume word -> raw;
        // the index of ray r + 2 and the word of ray r + 1 are in flight while ray r is composited
        const bool act = lane < St;
        struct Idx { int src; float zs, z1, dn; float4 re; };
        auto fetch_idx = [&](int r) {
            Idx in;
            const size_t m = (size_t)r * St + (act ? lane : St - 1);
            in.src = min(max(sorted_idx[m], 0), St - 1);     // NaN depths must not become an out-of-bounds read
            in.zs = z[m];
            in.z1 = (lane + 1 < St) ? z[m + 1] : 0.f;
            in.dn = ray_norm(rays_d, r);
            in.re = raw_empty ? raw_empty[r] : float4{0.f, 0.f, 0.f, 0.f};
            return in;
        };
        auto fetch_word = [&](int r, int src) -> uint32_t {
            if (src < S) return bits_a ? bits_a[(size_t)r * S + src] : 1u;
            return bits_b ? bits_b[(size_t)r * Sf + (src - S)] : 1u;
        };
        // item i of the launch: the i-th listed ray, or (no list) ray scattered_ray(i); the index of item i + 2 and the word of
        // item i + 1 are in flight while item i is composited
        const int n = ray_list ? min(max(*ray_count, 0), R) : R;
        auto ray_at = [&](int i) { return i < n ? (ray_list ? min(max(ray_list[i], 0), R - 1) : scattered_ray(i, scatter, R)) : -1; };
        int r = ray_at(wave), r_nxt = ray_at(wave + nwaves);
        if (r < 0) return;
        Idx cur = fetch_idx(r), nxt = fetch_idx(r_nxt >= 0 ? r_nxt : r);
        uint32_t cur_word = fetch_word(r, cur.src);
        for (int i = wave; i < n; i += nwaves) {
            CompositeState st = {1.0f, 0.f, 0.f, 0.f, 0.f, 0.f};
            const size_t m = (size_t)r * St + (act ? lane : St - 1);
            float4 rw = cur.re;
            if (cur_word != 0u) rw = cur.src < S ? raw_a[(size_t)r * S + cur.src] : raw_b[(size_t)r * Sf + (cur.src - S)];
            const int r_n = r_nxt >= 0 ? r_nxt : r;
            const uint32_t nxt_word = fetch_word(r_n, nxt.src);
            const int r_nn = ray_at(i + 2 * nwaves);
            const Idx nn = fetch_idx(r_nn >= 0 ? r_nn : r_n);
            const float gap = (lane + 1 < St) ? sub_rn(cur.z1, cur.zs) : 1e10f;
            float al, w;
            // a ray without an in-volume sample in either pass: constants, as in k_composite_importance below (bit for bit)
            const float dist = mul_rn(gap, cur.dn);
            const float rgb_sum = add_rn(add_rn(cur.re.x, cur.re.y), cur.re.z);
            const bool flat = raw_empty != nullptr && bits_a != nullptr && bits_b != nullptr && noise == nullptr &&
                              !(div_rn(cur.re.w, B) > 0.f) && sub_rn(rgb_sum, rgb_sum) == 0.f &&
                              __all(cur_word == 0u && sub_rn(dist, dist) == 0.f);
            if (flat) {
                al = 0.f;
                w = 0.f;
            } else {
                w = composite_chunk(st, rw, cur.zs, gap, cur.dn, B, noise != nullptr, noise ? noise[m] : 0.f, act, lane, al);
            }
            if (act) {
                if (weights) weights[m] = w;
                if (alpha_out) alpha_out[m] = al;
                if (raw_sorted) raw_sorted[m] = rw;
            }
            if (lane == 0) composite_finish(st, r, rgb_map, disp, acc_out);
            cur = nxt; cur_word = nxt_word; nxt = nn; r = r_n; r_nxt = r_nn;
        }
        return;
    }
    const int n_all = ray_list ? min(max(*ray_count, 0), R) : R;
    for (int i = wave; i < n_all; i += nwaves) {
        const int r = ray_list ? min(max(ray_list[i], 0), R - 1) : i;
        const float dn = ray_norm(rays_d, r);
        CompositeState st = {1.0f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int c0 = 0; c0 < St; c0 += 64) {
            const int s = c0 + lane;
            const bool act = s < St;
            const size_t m = (size_t)r * St + (act ? s : St - 1);
            const int src = min(max(sorted_idx[m], 0), St - 1);   // NaN depths must not become an out-of-bounds read
            float4 rw;
            if (src < S) {
                const size_t q = (size_t)r * S + src;
                rw = (bits_a && bits_a[q] == 0u) ? raw_empty[r] : raw_a[q];
            } else {
                const size_t q = (size_t)r * Sf + (src - S);
                rw = (bits_b && bits_b[q] == 0u) ? raw_empty[r] : raw_b[q];
            }
            const float zs = z[m];
            const float gap = (s + 1 < St) ? sub_rn(z[m + 1], zs) : 1e10f;
            float al;
            const float w = composite_chunk(st, rw, zs, gap, dn, B, noise != nullptr, noise ? noise[m] : 0.f, act, lane, al);
            if (act) {
                if (weights) weights[m] = w;
                if (alpha_out) alpha_out[m] = al;
                if (raw_sorted) raw_sorted[m] = rw;
            }
        }
        if (lane == 0) composite_finish(st, r, rgb_map, disp, acc_out);
    }
}

// ======================================================================================
// importance sampling + merge
// ======================================================================================
// General fallback (any S): one thread per ray, the sorted_idx row doubles as cdf scratch.
__global__ __launch_bounds__(64) void k_importance(const float* __restrict__ z, const float* __restrict__ weights,
                                                   int R, int S, int Sf, const float* __restrict__ u,
                                                   float* __restrict__ z_fine, float* __restrict__ z_sorted,
                                                   int32_t* __restrict__ sorted_idx) {
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < R; r += gridDim.x * blockDim.x) {
        float* cdf = reinterpret_cast<float*>(sorted_idx + (size_t)r * (S + Sf));
        const float* zr = z + (size_t)r * S;
        float* zf = z_fine + (size_t)r * Sf;
        float* zs = z_sorted + (size_t)r * (S + Sf);
        int32_t* si = sorted_idx + (size_t)r * (S + Sf);
        importance_ray(zr, weights + (size_t)r * S, S, Sf, u ? u + (size_t)r * Sf : nullptr, cdf, zf, zs, si);
        bool ascending = true;
        for (int i = 0; i + 1 < S; ++i) ascending = ascending && zr[i + 1] >= zr[i];
        if (u != nullptr || !ascending) {
            // random draws or descending depths (far bound before the near bound): the two-pointer merge above is not a
            // sort -> place every sample by its rank in (value, coarse-before-fine, index) order
            for (int i = 0; i < S; ++i) {
                int rank = 0;
                for (int k = 0; k < Sf; ++k) rank += zf[k] < zr[i];
                for (int q = 0; q < S; ++q) rank += (zr[q] < zr[i]) || (zr[q] == zr[i] && q < i);
                zs[rank] = zr[i];
                si[rank] = i;
            }
            for (int k = 0; k < Sf; ++k) {
                int rank = 0;
                for (int i = 0; i < S; ++i) rank += zr[i] <= zf[k];
                for (int q = 0; q < Sf; ++q) rank += (zf[q] < zf[k]) || (zf[q] == zf[k] && q < k);
                zs[rank] = zf[k];
                si[rank] = S + k;
            }
        }
    }
}

// S <= 64 and Sf <= 64: one wavefront per ray.  Lane i holds coarse sample i (pdf bin, cdf by a DPP scan);
// lane k additionally draws fine sample k (binary search in the cdf through shuffles); the merged order comes
// from ranks -- two more binary searches when the draws are the sorted linspace (eval), broadcast compares for
// random draws -- no sort, no scratch memory, every global access coalesced.
//   zi / wi: this lane's coarse depth (INFINITY beyond S) and weight (0 beyond S)
template <bool DET>
__device__ __forceinline__ void importance_wave(float zi, float wi, int r, int S, int Sf, const float* __restrict__ u,
                                                int lane, float* __restrict__ z_fine, float* __restrict__ z_sorted,
                                                int32_t* __restrict__ sorted_idx) {
    const int nb = S - 2, ncdf = S - 1;
    const bool cact = lane < S;
    const float w1 = __shfl_down(wi, 1, 64), w2 = __shfl_down(wi, 2, 64);
    const float z1 = __shfl_down(zi, 1, 64);
    float dw = 0.f;
    if (lane < nb) dw = add_rn(add_rn(mul_rn(0.5f, add_rn(fmaxf(wi, w1), fmaxf(w1, w2))), 0.01f), 1e-5f);
    const float sum = wave_total(dw);
    const float inc = wave_scan_add(div_rn(dw, sum));  // pdf -> inclusive scan
    float cdf = __shfl_up(inc, 1, 64);                 // lane i: cdf[i], i in [0, ncdf)
    if (lane == 0) cdf = 0.f;
    const float bin = mul_rn(0.5f, add_rn(z1, zi));    // lane i: mid-point i (valid for i < S-1)
    // ---- inverse CDF for fine sample `lane` ----
    const bool fact = lane < Sf;
    const float uk = DET ? linspace01(fact ? lane : 0, Sf) : (fact ? u[(size_t)r * Sf + lane] : 0.f);
    int lo = 0, hi = ncdf;  // searchsorted(cdf, u, right=True)
#pragma unroll
    for (int it = 0; it < 7; ++it) {
        const int mid = (lo + hi) >> 1;
        const float cm = __shfl(cdf, mid < ncdf ? mid : ncdf - 1, 64);
        if (lo < hi) {
            if (cm > uk) hi = mid; else lo = mid + 1;
        }
    }
    const int below = lo - 1 > 0 ? lo - 1 : 0;
    const int above = lo < ncdf - 1 ? lo : ncdf - 1;
    const float c0 = __shfl(cdf, below, 64), c1 = __shfl(cdf, above, 64);
    const float b0 = __shfl(bin, below, 64), b1 = __shfl(bin, above, 64);
    float denom = sub_rn(c1, c0);
    if (denom < 1e-5f) denom = 1.0f;
    const float t = div_rn(sub_rn(uk, c0), denom);
    const float zf = fact ? add_rn(b0, mul_rn(t, sub_rn(b1, b0))) : INFINITY;
    if (fact) z_fine[(size_t)r * Sf + lane] = zf;
    // ---- merged order by rank (stable: coarse first on ties, fine by index) ----
    int rank_c = lane, rank_f = lane;
    // the binary searches need non-decreasing depths; a ray whose far bound lies before its near bound (looking away
    // from the body) has descending samples -> count ranks the general way (wave-uniform choice)
    const bool ascending = __all(!(lane < S - 1) || z1 >= zi);
    if (DET && ascending) {
        // both sequences are non-decreasing: #fine < zi and #coarse <= zf by binary search
        int l0 = 0, h0 = Sf, l1 = 0, h1 = S;
#pragma unroll
        for (int it = 0; it < 7; ++it) {
            const int m0 = (l0 + h0) >> 1, m1 = (l1 + h1) >> 1;
            const float f = __shfl(zf, m0 < 63 ? m0 : 63, 64);
            const float c = __shfl(zi, m1 < 63 ? m1 : 63, 64);
            if (l0 < h0) { if (f < zi) l0 = m0 + 1; else h0 = m0; }
            if (l1 < h1) { if (c <= zf) l1 = m1 + 1; else h1 = m1; }
        }
        rank_c += l0;
        rank_f += l1;
    } else {
        // general case: rank = number of predecessors in (value, coarse-before-fine, index) order -- a permutation for
        // any input order, equal to torch.sort(cat([z, z_fine])) (stable)
        rank_c = 0;
        rank_f = 0;
        for (int k = 0; k < Sf; ++k) {
            const float zk = __shfl(zf, k, 64);
            rank_c += zk < zi;
            rank_f += (zk < zf) || (zk == zf && k < lane);
        }
        if (ascending) rank_c += lane;
        for (int i = 0; i < S; ++i) {
            const float zc = __shfl(zi, i, 64);
            rank_f += zc <= zf;
            if (!ascending) rank_c += (zc < zi) || (zc == zi && i < lane);
        }
    }
    const size_t o = (size_t)r * (S + Sf);
    if (cact) { z_sorted[o + rank_c] = zi; sorted_idx[o + rank_c] = lane; }
    if (fact) { z_sorted[o + rank_f] = zf; sorted_idx[o + rank_f] = S + lane; }
}

template <bool DET>
__global__ __launch_bounds__(256) void k_importance_wave(const float* __restrict__ z, const float* __restrict__ weights,
                                                         int R, int S, int Sf, const float* __restrict__ u,
                                                         float* __restrict__ z_fine, float* __restrict__ z_sorted,
                                                         int32_t* __restrict__ sorted_idx) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    for (int r = wave; r < R; r += nwaves) {
        const bool cact = lane < S;
        const float zi = cact ? z[(size_t)r * S + lane] : INFINITY;
        const float wi = cact ? weights[(size_t)r * S + lane] : 0.f;
        importance_wave<DET>(zi, wi, r, S, Sf, u, lane, z_fine, z_sorted, sorted_idx);
    }
}

// 64 < S <= 256, Sf <= 64 (config 3: 96 + 32): one wavefront per ray, the coarse samples in chunks of 64, depths / weights / cdf
// in 3 KB of LDS per wavefront.  (These shapes used to take k_importance's thread-per-ray walk: 2.85 ms of config 3's 11.7 ms
// frame.)  Same arithmetic as importance_wave: pdf, total and cdf by DPP scans (chunk after chunk, the carry added per chunk),
// inverse CDF and ranks by binary searches -- in LDS instead of shuffles.
constexpr int IMPB_MAX_S = 256;
template <bool DET>
__global__ __launch_bounds__(256) void k_importance_wave_long(const float* __restrict__ z, const float* __restrict__ weights,
                                                              int R, int S, int Sf, const float* __restrict__ u,
                                                              float* __restrict__ z_fine, float* __restrict__ z_sorted,
                                                              int32_t* __restrict__ sorted_idx, unsigned scatter,
                                                              const int32_t* __restrict__ ray_list,
                                                              const int32_t* __restrict__ ray_count) {
    __shared__ float s_all[4][3 * IMPB_MAX_S + 64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float* s_z = s_all[wv];                   // [S] coarse depths
    float* s_w = s_z + IMPB_MAX_S;            // [S] weights, then [S - 1] the cdf
    float* s_b = s_w + IMPB_MAX_S;            // [S - 1] mid-points
    float* s_f = s_b + IMPB_MAX_S;            // [64] importance depths
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    const int nb = S - 2, ncdf = S - 1;
    const int nchunk = (S + 63) >> 6;
    const int n = ray_list ? min(max(*ray_count, 0), R) : R;
    for (int i = wave; i < n; i += nwaves) {
        const int r = ray_list ? min(max(ray_list[i], 0), R - 1) : scattered_ray(i, scatter, R);
        const size_t o = (size_t)r * S;
        for (int c = 0; c < nchunk; ++c) {
            const int s0 = 64 * c + lane;
            if (s0 < S) { s_z[s0] = z[o + s0]; s_w[s0] = weights[o + s0]; }
        }
        // (one wavefront: its LDS writes are complete before its next LDS reads -- the compiler's lgkmcnt waits)
        float sum = 0.f;
        float dwv[IMPB_MAX_S / 64];
        bool asc = true;
#pragma unroll
        for (int c = 0; c < IMPB_MAX_S / 64; ++c) {
            const int s0 = 64 * c + lane;
            float dw = 0.f;
            if (c < nchunk && s0 < nb) {
                const float w0 = s_w[s0], w1 = s_w[s0 + 1], w2 = s_w[s0 + 2];
                dw = add_rn(add_rn(mul_rn(0.5f, add_rn(fmaxf(w0, w1), fmaxf(w1, w2))), 0.01f), 1e-5f);
            }
            dwv[c] = dw;
            if (c < nchunk) {
                sum = add_rn(sum, wave_total(dw));
                if (s0 + 1 < S) {
                    const float z0 = s_z[s0], z1 = s_z[s0 + 1];
                    s_b[s0] = mul_rn(0.5f, add_rn(z1, z0));
                    asc = asc && z1 >= z0;
                }
            }
        }
        const bool ascending = __all(asc);
        float carry = 0.f;
#pragma unroll
        for (int c = 0; c < IMPB_MAX_S / 64; ++c) {
            if (c < nchunk) {
                const int s0 = 64 * c + lane;
                const float inc = add_rn(carry, wave_scan_add(div_rn(dwv[c], sum)));     // cdf[s0 + 1]
                if (s0 < nb) s_w[s0 + 1] = inc;
                carry = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, inc), 63));
            }
        }
        if (lane == 0) s_w[0] = 0.f;
        // ---- inverse CDF for fine sample `lane` ----
        const bool fact = lane < Sf;
        const float uk = DET ? linspace01(fact ? lane : 0, Sf) : (fact ? u[(size_t)r * Sf + lane] : 0.f);
        int lo = 0, hi = ncdf;  // searchsorted(cdf, u, right=True)
#pragma unroll
        for (int it = 0; it < 9; ++it) {
            const int mid = (lo + hi) >> 1;
            const float cm = s_w[mid < ncdf ? mid : ncdf - 1];
            if (lo < hi) {
                if (cm > uk) hi = mid; else lo = mid + 1;
            }
        }
        const int below = lo - 1 > 0 ? lo - 1 : 0;
        const int above = lo < ncdf - 1 ? lo : ncdf - 1;
        const float c0 = s_w[below], c1 = s_w[above];
        const float b0 = s_b[below], b1 = s_b[above];
        float denom = sub_rn(c1, c0);
        if (denom < 1e-5f) denom = 1.0f;
        const float t = div_rn(sub_rn(uk, c0), denom);
        const float zf = fact ? add_rn(b0, mul_rn(t, sub_rn(b1, b0))) : INFINITY;
        if (fact) z_fine[(size_t)r * Sf + lane] = zf;
        s_f[lane] = zf;
        // ---- merged order by rank (stable: coarse first on ties, fine by index) ----
        const size_t oo = (size_t)r * (S + Sf);
        if (DET && ascending) {
            // both sequences are non-decreasing: #coarse <= zf and #fine < z_i by binary search
            int l1 = 0, h1 = S;
#pragma unroll
            for (int it = 0; it < 9; ++it) {
                const int m1 = (l1 + h1) >> 1;
                const float cz = s_z[m1 < S ? m1 : S - 1];
                if (l1 < h1) { if (cz <= zf) l1 = m1 + 1; else h1 = m1; }
            }
            if (fact) { z_sorted[oo + lane + l1] = zf; sorted_idx[oo + lane + l1] = S + lane; }
            for (int c = 0; c < nchunk; ++c) {
                const int s0 = 64 * c + lane;
                const float zi = s_z[s0 < S ? s0 : S - 1];
                int l0 = 0, h0 = Sf;
#pragma unroll
                for (int it = 0; it < 7; ++it) {
                    const int m0 = (l0 + h0) >> 1;
                    const float fz = s_f[m0 < 63 ? m0 : 63];
                    if (l0 < h0) { if (fz < zi) l0 = m0 + 1; else h0 = m0; }
                }
                if (s0 < S) { z_sorted[oo + s0 + l0] = zi; sorted_idx[oo + s0 + l0] = s0; }
            }
        } else {
            // general case: rank = number of predecessors in (value, coarse-before-fine, index) order
            int rank_f = 0;
            for (int k = 0; k < Sf; ++k) {
                const float zk = s_f[k];
                rank_f += (zk < zf) || (zk == zf && k < lane);
            }
            for (int q = 0; q < S; ++q) rank_f += s_z[q] <= zf;
            if (fact) { z_sorted[oo + rank_f] = zf; sorted_idx[oo + rank_f] = S + lane; }
            for (int c = 0; c < nchunk; ++c) {
                const int s0 = 64 * c + lane;
                const float zi = s_z[s0 < S ? s0 : S - 1];
                int rank_c = 0;
                for (int k = 0; k < Sf; ++k) rank_c += s_f[k] < zi;
                if (ascending) rank_c += s0;
                else
                    for (int q = 0; q < S; ++q) { const float zq = s_z[q]; rank_c += (zq < zi) || (zq == zi && q < s0); }
                if (s0 < S) { z_sorted[oo + rank_c] = zi; sorted_idx[oo + rank_c] = s0; }
            }
        }
    }
}

// coarse composite + importance resampling of a ray in one pass (S, Sf <= 64): the weights never leave the
// wavefront's registers unless the caller asks for them.  Item i of the launch is the i-th listed ray (ray_list / ray_count:
// k_flat_rays' list of the rays that are NOT rays of constants) or, without a list, ray scattered_ray(i).
template <bool DET>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_composite_importance(const float4* __restrict__ raw,
                                                              const float4* __restrict__ raw_empty,
                                                              const uint32_t* __restrict__ bits,
                                                              const float* __restrict__ z, const float* __restrict__ rays_d,
                                                              int R, int S, int Sf, float B, const float* __restrict__ noise,
                                                              const float* __restrict__ u, float* __restrict__ rgb_map,
                                                              float* __restrict__ disp, float* __restrict__ acc_out,
                                                              float* __restrict__ weights, float* __restrict__ alpha_out,
                                                              float* __restrict__ z_fine, float* __restrict__ z_sorted,
                                                              int32_t* __restrict__ sorted_idx,
                                                              const int32_t* __restrict__ ray_list,
                                                              const int32_t* __restrict__ ray_count, unsigned scatter) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    const bool act = lane < S;
    // A wavefront walks its rays one after the other and a ray is a chain of dependent loads (list entry -> in-volume word -> raw)
    // before any arithmetic: software-pipelined -- the next ray's word, depths, direction and empty-space raw (and the list entry
    // after it) are requested before this ray's arithmetic, so only the (masked) raw load is waited for at full latency
    struct RayIn { uint32_t word; float zs, z1, dn; float4 re; };
    auto fetch = [&](int r) {
        RayIn in;
        const size_t m = (size_t)r * S + (act ? lane : S - 1);
        in.word = bits ? bits[m] : 1u;
        in.zs = z[m];
        in.z1 = (lane + 1 < S) ? z[m + 1] : 0.f;
        in.dn = ray_norm(rays_d, r);
        in.re = bits ? raw_empty[r] : float4{0.f, 0.f, 0.f, 0.f};
        return in;
    };
    const int n = ray_list ? min(max(*ray_count, 0), R) : R;
    auto ray_at = [&](int i) { return i < n ? (ray_list ? min(max(ray_list[i], 0), R - 1) : scattered_ray(i, scatter, R)) : -1; };
    int r = ray_at(wave), r_nxt = ray_at(wave + nwaves);
    if (r < 0) return;
    RayIn cur = fetch(r);
    for (int i = wave; i < n; i += nwaves) {
        CompositeState st = {1.0f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const size_t m = (size_t)r * S + (act ? lane : S - 1);
        float4 rw = cur.re;
        if (cur.word != 0u) rw = raw[m];
        const int r_n = r_nxt >= 0 ? r_nxt : r;
        const RayIn nxt = fetch(r_n);
        const int r_nn = ray_at(i + 2 * nwaves);
        const float zs = cur.zs;
        const float gap = (lane + 1 < S) ? sub_rn(cur.z1, zs) : 1e10f;
        float al, w;
        // A ray with no sample inside any volume carries ONE raw for all its samples -- the ray's empty-space raw.  If its density
        // pre-activation is <= 0 (or NaN: fmaxf drops it) and every interval length is finite, the general chain below computes
        // sig = 0, alpha = 1 - exp(-0) = +0, T = 1, w = +0 for every sample and +0 for all five sums: the maps are constants.
        // Taken wave-uniformly, bit for bit the general result (the importance depths still go through importance_wave: they
        // depend on the ray's depths only).
        const float dist = mul_rn(gap, cur.dn);
        const float rgb_sum = add_rn(add_rn(cur.re.x, cur.re.y), cur.re.z);     // (NaN / inf colour logits would make 0 * c a NaN)
        const bool flat = bits != nullptr && noise == nullptr && !(div_rn(cur.re.w, B) > 0.f) && sub_rn(rgb_sum, rgb_sum) == 0.f &&
                          __all(cur.word == 0u && sub_rn(dist, dist) == 0.f);
        if (flat) {
            al = 0.f;
            w = 0.f;
            st.sr = st.sg = st.sb = st.sd = st.sa = 0.f;
        } else {
            w = composite_chunk(st, rw, zs, gap, cur.dn, B, noise != nullptr, noise ? noise[m] : 0.f, act, lane, al);
        }
        if (act) {
            if (weights) weights[m] = w;
            if (alpha_out) alpha_out[m] = al;
        }
        if (lane == 0) composite_finish(st, r, rgb_map, disp, acc_out);
        importance_wave<DET>(act ? zs : INFINITY, act ? w : 0.f, r, S, Sf, u, lane, z_fine, z_sorted, sorted_idx);
        cur = nxt; r = r_n; r_nxt = r_nn;
    }
}

// Rays of constants.  A ray that cannot meet a volume anywhere between t_lo and t_hi (k_ray_bone_mask's flag), all of whose coarse
// depths lie in that interval (k_bone_cull clears the flag otherwise), has ONE raw on every sample of BOTH passes -- its
// importance depths would lie between its coarse depths -- the ray's empty-space raw.  If the model's empty-space density is <= 0
// and its empty-space colour logits cannot be NaN (the CALLER's statement about the model, danbo_hip.h: both are properties of
// the weights, not of the ray), that is sig = 0, alpha = +0, T = 1, w = +0 everywhere.  This kernel writes every output of the
// two composites for those rays (the values the general chain computes, bit for bit: +0 everywhere), z_fine = t_lo (inside the
// interval: the importance pass's cull drops the ray on its mask), and lists all OTHER rays for k_view_consts,
// k_composite_importance and k_composite_merged -- which then share them evenly over their wavefronts (a static split of ALL
// rays left the wavefronts with 5 to 25 rays of work each: as slow as without the flags).  63 % of the rays of the bench frame.
constexpr int FLAT_BLOCK = 256;   // (small workgroups: the launch runs beside the cull and has to fit into the wave slots that one leaves)
__global__ __launch_bounds__(FLAT_BLOCK) void k_flat_rays(const float* __restrict__ t_lo,
                                                          const uint32_t* __restrict__ ray_flat, int R, int S, int Sf,
                                                          float* __restrict__ rgb0, float* __restrict__ disp0, float* __restrict__ acc0,
                                                          float* __restrict__ weights0, float* __restrict__ alpha0,
                                                          float* __restrict__ z_fine, float* __restrict__ rgb_map,
                                                          float* __restrict__ disp, float* __restrict__ acc_out,
                                                          float* __restrict__ weights, float* __restrict__ alpha_out,
                                                          int32_t* __restrict__ ray_list, int32_t* __restrict__ ray_count,
                                                          int parts) {
    __shared__ int s_cnt[FLAT_BLOCK / 64];
    __shared__ int s_base;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int St = S + Sf;
    const int r = blockIdx.x * FLAT_BLOCK + threadIdx.x;
    const bool flat = r < R && ray_flat[r] != 0u;
    const float lo = flat ? t_lo[r] : 0.f;
    if (flat && (parts & 1)) {     // composite_finish of five +0 sums, twice
        rgb0[3 * r] = 0.f; rgb0[3 * r + 1] = 0.f; rgb0[3 * r + 2] = 0.f;
        disp0[r] = 0.f;
        acc0[r] = 0.f;
        rgb_map[3 * r] = 0.f; rgb_map[3 * r + 1] = 0.f; rgb_map[3 * r + 2] = 0.f;
        disp[r] = 0.f;
        acc_out[r] = 0.f;
    }
    // the rows of the wavefront's flat rays, a ray per turn
    unsigned long long rows = (parts & 2) ? __ballot(flat) : 0ull;
    const unsigned long long listed = __ballot(r < R && !flat);
    while (rows != 0ull) {
        const int b = (int)__builtin_ctzll(rows);
        rows &= rows - 1ull;
        const size_t rr = (size_t)(blockIdx.x * FLAT_BLOCK + wave * 64 + b);
        for (int c = lane; c < S; c += 64) {
            if (weights0) weights0[rr * S + c] = 0.f;
            if (alpha0) alpha0[rr * S + c] = 0.f;
        }
        const float lo_b = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, lo), b));
        for (int c = lane; c < Sf; c += 64) z_fine[rr * Sf + c] = lo_b;
        for (int c = lane; c < St; c += 64) {
            if (weights) weights[rr * St + c] = 0.f;
            if (alpha_out) alpha_out[rr * St + c] = 0.f;
        }
    }
    if (!(parts & 1)) return;
    // the other rays, in ray order inside the workgroup: one atomic per workgroup
    if (lane == 0) s_cnt[wave] = (int)__popcll(listed);
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int w = 0; w < FLAT_BLOCK / 64; ++w) { const int c = s_cnt[w]; s_cnt[w] = run; run += c; }
        s_base = run > 0 ? atomicAdd(ray_count, run) : 0;
    }
    __syncthreads();
    if (r < R && !flat) ray_list[s_base + s_cnt[wave] + (int)__popcll(listed & ((1ull << lane) - 1ull))] = r;
}

}  // namespace danbo

// ======================================================================================
// C ABI
// ======================================================================================
using namespace danbo;

// a multiplier coprime to R, far from the image widths (scattered_ray)
static unsigned ray_scatter(int R) {
    static const unsigned primes[] = {40507u, 40519u, 40529u, 40531u, 40543u, 40559u};
    for (unsigned p : primes) {
        unsigned a = p, b = (unsigned)R;
        while (b != 0u) { const unsigned t = a % b; a = b; b = t; }
        if (a == 1u) return p;
    }
    return 1u;
}

extern "C" int danbo_near_far_cylinder(const float* rays_o, const float* rays_d, const float* cyl, int R, int G,
                                        float near0, float far0, const float* near_in, const float* far_in, int chunk,
                                        float* scratch, float* near_out, float* far_out, void* stream) {
    DANBO_CHECK_ARG(R > 0 && G > 0 && R % G == 0 && chunk > 0 && scratch != nullptr);
    hipStream_t st = (hipStream_t)stream;
    const int nchunk = ceil_div(R, chunk);
    if (chunk <= CYL_FUSED_MAX) {       // (the scratch stays untouched)
        hipLaunchKernelGGL(k_cylinder_chunk, dim3(nchunk), dim3(CYL_BLOCK), 0, st, rays_o, rays_d, cyl, R, G, near0, far0, near_in,
                           far_in, chunk, near_out, far_out);
        DANBO_LAUNCH_RET();
    }
    zero_words(scratch, 8L * nchunk, nullptr, 0, st);
    const int grid = stream_grid(R, 256);
    hipLaunchKernelGGL(k_cylinder_pass1, dim3(grid), dim3(256), 0, st, rays_o, rays_d, cyl, R, G, near0, far0, near_in,
                       far_in, chunk, reinterpret_cast<double*>(scratch), near_out, far_out);
    hipLaunchKernelGGL(k_cylinder_pass2, dim3(grid), dim3(256), 0, st, R, near0, far0, near_in, far_in, chunk,
                       reinterpret_cast<const double*>(scratch), near_out, far_out);
    DANBO_LAUNCH_RET();
}

extern "C" int danbo_near_far_boxes(const float* rays_o, const float* rays_d, const float* skts, const float* align,
                                     const float* axis_scale, int R, int G, float* near_io, float* far_io,
                                     void* stream) {
    DANBO_CHECK_ARG(R > 0 && G > 0 && R % G == 0);
    hipLaunchKernelGGL(k_box_bounds, dim3(stream_grid((long)R * 32, 256)), dim3(256), 0, (hipStream_t)stream, rays_o, rays_d, skts,
                       align, axis_scale, R, G, near_io, far_io);
    DANBO_LAUNCH_RET();
}

extern "C" int danbo_coarse_samples(const float* near, const float* far, int R, int S, const float* t_rand, float* z,
                                     void* stream) {
    DANBO_CHECK_ARG(R > 0 && S > 0);
    if (S % 4 == 0 && (long)R * S <= 0x7fffffffL && ((uintptr_t)z & 15) == 0)
        hipLaunchKernelGGL(k_coarse_samples4, dim3(stream_grid((long)R * S / 4, 256)), dim3(256), 0, (hipStream_t)stream, near, far, R, S,
                           t_rand, reinterpret_cast<float4*>(z));
    else
        hipLaunchKernelGGL(k_coarse_samples, dim3(stream_grid((long)R * S, 256)), dim3(256), 0, (hipStream_t)stream, near,
                           far, R, S, t_rand, z);
    DANBO_LAUNCH_RET();
}

extern "C" int danbo_bone_cull(const float* rays_o, const float* rays_d, const float* z, const float* pts, int R, int S, int G,
                                const float* skts, const float* align, const float* axis_scale, const uint32_t* ray_mask,
                                const float* t_lo, const float* t_hi, uint32_t* ray_flat, uint32_t* valid_bits, int32_t* list,
                                int32_t* count, void* stream) {
    DANBO_CHECK_ARG(R > 0 && S > 0 && G > 0 && R % G == 0);
    DANBO_CHECK_ARG((list == nullptr) == (count == nullptr));
    DANBO_CHECK_ARG((z == nullptr) != (pts == nullptr));
    DANBO_CHECK_ARG(ray_mask == nullptr || (z != nullptr && t_lo != nullptr && t_hi != nullptr));
    DANBO_CHECK_ARG(ray_flat == nullptr || ray_mask != nullptr);
    const long M = (long)R * S;
    const long spp = (long)(R / G) * S;
    const int per_block = CULL_BLOCK * CULL_SPT;
    long np = per_block / spp + 2;
    if (np > G) np = G;
    const size_t lds = sizeof(float) * (J * 16 + J * 4 + np * J * 16);
    DANBO_CHECK_ARG(lds <= 64 * 1024);
    const int grid = ceil_div(M, per_block);
    hipLaunchKernelGGL(k_bone_cull, dim3(grid), dim3(CULL_BLOCK), lds, (hipStream_t)stream, rays_o, rays_d, z, pts, R, S,
                       G, skts, align, axis_scale, (int)np, ray_mask, t_lo, t_hi, ray_flat, valid_bits, list, count);
    DANBO_LAUNCH_RET();
}

extern "C" int danbo_ray_bone_mask(const float* rays_o, const float* rays_d, const float* t_lo, const float* t_hi, int R, int G,
                                    const float* skts, const float* align, const float* axis_scale, uint32_t* ray_mask,
                                    uint32_t* ray_flat, void* stream) {
    DANBO_CHECK_ARG(rays_o && rays_d && t_lo && t_hi && skts && align && axis_scale && ray_mask);
    DANBO_CHECK_ARG(R > 0 && G > 0 && R % G == 0);
    const long np = 256 / (R / G) + 2 < G ? 256 / (R / G) + 2 : G;     // poses a workgroup's 256 rays can span
    const bool in_lds = np <= 8;
    const size_t lds = sizeof(float) * (J * 16 + J * 4 + (in_lds ? np * J * 16 : 0));
    if (in_lds)
        hipLaunchKernelGGL(k_ray_bone_mask<true>, dim3(ceil_div(R, 256)), dim3(256), lds, (hipStream_t)stream, rays_o, rays_d, t_lo, t_hi,
                           skts, align, axis_scale, R, G, ray_mask, ray_flat);
    else
        hipLaunchKernelGGL(k_ray_bone_mask<false>, dim3(ceil_div(R, 256)), dim3(256), lds, (hipStream_t)stream, rays_o, rays_d, t_lo,
                           t_hi, skts, align, axis_scale, R, G, ray_mask, ray_flat);
    DANBO_LAUNCH_RET();
}

extern "C" int danbo_bone_gather_fwd(const float* rays_o, const float* rays_d, const float* z, const float* pts, int R, int S, int G,
                                      const float* skts, const float* align, const float* axis_scale,
                                      const float* volumes, const int32_t* list, const int32_t* count, int n,
                                      float* part_feat, void* stream) {
    DANBO_CHECK_ARG(R > 0 && S > 0 && G > 0 && R % G == 0 && n >= 0);
    DANBO_CHECK_ARG((z == nullptr) != (pts == nullptr));
    if (n == 0) return 0;
    const int ntiles = ceil_div(n, GATHER_TS);
    const int grid = ntiles < num_cu() * 3 ? ntiles : num_cu() * 3;
    hipLaunchKernelGGL(k_bone_gather, dim3(grid), dim3(GATHER_BLOCK), 0, (hipStream_t)stream, rays_o, rays_d, z, pts, R, S,
                       G, skts, align, axis_scale, volumes, list, count, n, part_feat);
    DANBO_LAUNCH_RET();
}

extern "C" int danbo_fill_raw(const float* raw_empty, int R, int S, float* raw, void* stream) {
    DANBO_CHECK_ARG(R > 0 && S > 0);
    hipLaunchKernelGGL(k_fill_raw, dim3(stream_grid((long)R * S, 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4*>(raw_empty), R, S, reinterpret_cast<float4*>(raw));
    DANBO_LAUNCH_RET();
}

extern "C" int danbo_merge_samples(const float* a, const float* b, const int32_t* sorted_idx, int R, int S, int Sf,
                                    int C, float* out, void* stream) {
    DANBO_CHECK_ARG(R > 0 && S > 0 && Sf >= 0 && C > 0);
    hipLaunchKernelGGL(k_merge_samples, dim3(stream_grid((long)R * (S + Sf), 256)), dim3(256), 0, (hipStream_t)stream, a,
                       b, sorted_idx, R, S, Sf, C, out);
    DANBO_LAUNCH_RET();
}

static int composite_impl(const float* raw, const float* raw_empty, const uint32_t* bits, const float* z, const float* rays_d, int R,
                          int S, float B, const float* noise, float* rgb_map, float* disp, float* acc, float* weights, float* alpha,
                          const int32_t* ray_list, const int32_t* ray_count, void* stream) {
    DANBO_CHECK_ARG(R > 0 && S > 0 && B > 0.f);
    DANBO_CHECK_ARG((ray_list == nullptr) == (ray_count == nullptr) && (bits == nullptr || raw_empty != nullptr));
    static const int per_launch = resident_grid(k_composite, 1L << 40, 256);
    const int grid = (int)std::min<long>(ceil_div((long)R * 64, 256), per_launch);
    hipLaunchKernelGGL(k_composite, dim3(grid), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const float4*>(raw),
                       reinterpret_cast<const float4*>(raw_empty), bits, z, rays_d, R, S, B, noise, rgb_map, disp, acc, weights, alpha,
                       ray_list, ray_count);
    DANBO_LAUNCH_RET();
}
extern "C" int danbo_composite_fwd(const float* raw, const float* z, const float* rays_d, int R, int S, float B,
                                    const float* noise, float* rgb_map, float* disp, float* acc, float* weights,
                                    float* alpha, void* stream) {
    return composite_impl(raw, nullptr, nullptr, z, rays_d, R, S, B, noise, rgb_map, disp, acc, weights, alpha, nullptr, nullptr, stream);
}
extern "C" int danbo_composite_rays_fwd(const float* raw, const float* raw_empty, const uint32_t* valid_bits, const float* z,
                                         const float* rays_d, int R, int S, float B, const float* noise, float* rgb_map,
                                         float* disp, float* acc, float* weights, float* alpha, const int32_t* ray_list,
                                         const int32_t* ray_count, void* stream) {
    return composite_impl(raw, raw_empty, valid_bits, z, rays_d, R, S, B, noise, rgb_map, disp, acc, weights, alpha, ray_list, ray_count,
                          stream);
}

static int importance_impl(const float* z, const float* weights, int R, int S, int Sf, const float* u, float* z_fine,
                           float* z_sorted, int32_t* sorted_idx, const int32_t* ray_list, const int32_t* ray_count, void* stream) {
    DANBO_CHECK_ARG(R > 0 && S >= 3 && Sf > 0);
    DANBO_CHECK_ARG((ray_list == nullptr) == (ray_count == nullptr));
    DANBO_CHECK_ARG(ray_list == nullptr || (S > 64 && S <= IMPB_MAX_S && Sf <= 64));      // (the list: the long-ray kernel only)
    if (S <= 64 && Sf <= 64) {
        if (u)
            hipLaunchKernelGGL(k_importance_wave<false>, dim3(stream_grid((long)R * 64, 256)), dim3(256), 0,
                               (hipStream_t)stream, z, weights, R, S, Sf, u, z_fine, z_sorted, sorted_idx);
        else
            hipLaunchKernelGGL(k_importance_wave<true>, dim3(stream_grid((long)R * 64, 256)), dim3(256), 0,
                               (hipStream_t)stream, z, weights, R, S, Sf, u, z_fine, z_sorted, sorted_idx);
    } else if (S <= IMPB_MAX_S && Sf <= 64) {
        const dim3 grid(stream_grid((long)R * 64, 256));
        if (u)
            hipLaunchKernelGGL(k_importance_wave_long<false>, grid, dim3(256), 0, (hipStream_t)stream, z, weights, R, S, Sf, u, z_fine,
                               z_sorted, sorted_idx, ray_scatter(R), ray_list, ray_count);
        else
            hipLaunchKernelGGL(k_importance_wave_long<true>, grid, dim3(256), 0, (hipStream_t)stream, z, weights, R, S, Sf, u, z_fine,
                               z_sorted, sorted_idx, ray_scatter(R), ray_list, ray_count);
    } else {
        hipLaunchKernelGGL(k_importance, dim3(stream_grid(R, 64)), dim3(64), 0, (hipStream_t)stream, z, weights, R, S, Sf,
                           u, z_fine, z_sorted, sorted_idx);
    }
    DANBO_LAUNCH_RET();
}
extern "C" int danbo_importance_samples(const float* z, const float* weights, int R, int S, int Sf, const float* u,
                                         float* z_fine, float* z_sorted, int32_t* sorted_idx, void* stream) {
    return importance_impl(z, weights, R, S, Sf, u, z_fine, z_sorted, sorted_idx, nullptr, nullptr, stream);
}
extern "C" int danbo_importance_samples_rays(const float* z, const float* weights, int R, int S, int Sf, const float* u,
                                              float* z_fine, float* z_sorted, int32_t* sorted_idx, const int32_t* ray_list,
                                              const int32_t* ray_count, void* stream) {
    return importance_impl(z, weights, R, S, Sf, u, z_fine, z_sorted, sorted_idx, ray_list, ray_count, stream);
}

extern "C" int danbo_composite_importance_fwd(const float* raw, const float* raw_empty, const uint32_t* valid_bits,
                                               const float* z, const float* rays_d, int R, int S, int Sf, float B,
                                               const float* noise, const float* u, float* rgb_map, float* disp,
                                               float* acc, float* weights, float* alpha, float* z_fine, float* z_sorted,
                                               int32_t* sorted_idx, const int32_t* ray_list, const int32_t* ray_count,
                                               void* stream) {
    DANBO_CHECK_ARG(R > 0 && S >= 3 && S <= 64 && Sf > 0 && Sf <= 64 && B > 0.f);
    DANBO_CHECK_ARG(raw && z && rays_d && rgb_map && disp && acc && z_fine && z_sorted && sorted_idx);
    DANBO_CHECK_ARG((valid_bits == nullptr) || (raw_empty != nullptr));
    DANBO_CHECK_ARG((ray_list == nullptr) == (ray_count == nullptr));
    static const int resident[2] = {resident_grid(k_composite_importance<false>, 1L << 40, 256), resident_grid(k_composite_importance<true>, 1L << 40, 256)};
    const dim3 grid((unsigned)std::min<long>(ceil_div((long)R * 64, 256), resident[u ? 0 : 1])), block(256);
    const float4* r4 = reinterpret_cast<const float4*>(raw);
    const float4* e4 = reinterpret_cast<const float4*>(raw_empty);
    if (u)
        hipLaunchKernelGGL(k_composite_importance<false>, grid, block, 0, (hipStream_t)stream, r4, e4, valid_bits, z, rays_d,
                           R, S, Sf, B, noise, u, rgb_map, disp, acc, weights, alpha, z_fine, z_sorted, sorted_idx, ray_list, ray_count,
                           ray_scatter(R));
    else
        hipLaunchKernelGGL(k_composite_importance<true>, grid, block, 0, (hipStream_t)stream, r4, e4, valid_bits, z, rays_d,
                           R, S, Sf, B, noise, u, rgb_map, disp, acc, weights, alpha, z_fine, z_sorted, sorted_idx, ray_list, ray_count,
                           ray_scatter(R));
    DANBO_LAUNCH_RET();
}

extern "C" int danbo_flat_rays(const float* t_lo, const uint32_t* ray_flat, int R, int S, int Sf, float* rgb0, float* disp0, float* acc0, float* weights0, float* alpha0, float* z_fine,
                                float* rgb_map, float* disp, float* acc, float* weights, float* alpha, int32_t* ray_list,
                                int32_t* ray_count, int parts, void* stream) {
    DANBO_CHECK_ARG(R > 0 && S >= 3 && Sf > 0);
    DANBO_CHECK_ARG(t_lo && ray_flat && rgb0 && disp0 && acc0 && z_fine && rgb_map && disp && acc && ray_list && ray_count);
    DANBO_CHECK_ARG(parts >= 1 && parts <= 3);
    hipLaunchKernelGGL(k_flat_rays, dim3(ceil_div(R, FLAT_BLOCK)), dim3(FLAT_BLOCK), 0, (hipStream_t)stream,
                       t_lo, ray_flat, R, S, Sf, rgb0, disp0, acc0, weights0, alpha0, z_fine,
                       rgb_map, disp, acc, weights, alpha, ray_list, ray_count, parts);
    DANBO_LAUNCH_RET();
}

extern "C" int danbo_composite_merged_fwd(const float* raw_a, const float* raw_b, const float* raw_empty,
                                           const uint32_t* bits_a, const uint32_t* bits_b, const int32_t* sorted_idx,
                                           const float* z_sorted, const float* rays_d, int R, int S, int Sf, float B,
                                           const float* noise, float* rgb_map, float* disp, float* acc, float* weights,
                                           float* alpha, float* raw_sorted, const int32_t* ray_list, const int32_t* ray_count,
                                           void* stream) {
    DANBO_CHECK_ARG(R > 0 && S > 0 && Sf > 0 && B > 0.f && raw_a && raw_b && sorted_idx && z_sorted && rays_d);
    DANBO_CHECK_ARG(rgb_map && disp && acc && ((bits_a == nullptr && bits_b == nullptr) || raw_empty != nullptr));
    DANBO_CHECK_ARG((ray_list == nullptr) == (ray_count == nullptr));
    static const int per_launch = resident_grid(k_composite_merged, 1L << 40, 256);
    hipLaunchKernelGGL(k_composite_merged, dim3((unsigned)std::min<long>(ceil_div((long)R * 64, 256), per_launch)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4*>(raw_a), reinterpret_cast<const float4*>(raw_b),
                       reinterpret_cast<const float4*>(raw_empty), bits_a, bits_b, sorted_idx, z_sorted, rays_d, R, S, Sf, B,
                       noise, rgb_map, disp, acc, weights, alpha, reinterpret_cast<float4*>(raw_sorted), ray_list, ray_count, ray_scatter(R));
    DANBO_LAUNCH_RET();
}
