// K3: positional encoding + density/colour MLP of DANBO (D=8, W=256, skip after layer 4,
// view_W=128) as ONE persistent kernel on the fp32 matrix cores (v_mfma_f32_32x32x2_f32, exact
// fp32 == an fmaf chain; MI355X peak 157 TFLOP/s).  gfx950 only.
//
// Workgroup = 256 threads = 4 wavefronts, one 64-row tile at a time:
//   * activations stay in LDS for all 11 GEMMs: X0 = PE(h) [64 x 200] (kept for the skip
//     connection) and ACT [64 x 256]; row strides 204 / 260 floats make both the MFMA A-fragment
//     ds_read_b128 (lane = row) and the epilogue ds_write_b32 (lane = column) conflict-free;
//   * each wavefront owns 64 (view layer: 32) output columns and streams its weights straight
//     from L2 into registers -- weights are pre-packed by danbo_mlp_pack into fragment order so a
//     wavefront's load is one contiguous 1 KB request ([col-tile][k-chunk of 8][lane][4]);
//     2.6 MB of packed weights stay L2-resident (4 MB L2 per XCD);
//   * per k-chunk: 2 ds_read_b128 (A, two row tiles) + 2 global_load_dwordx4 (B, two column
//     tiles) feed 16 MFMAs (1024 cycles of matrix pipe per wavefront);
//   * the per-ray part of the view layer arrives pre-reduced in `cview` (danbo_view_consts).
#include "common.hpp"

namespace danbo {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int MLP_BM = 64;
constexpr int MLP_W = 256;
constexpr int MLP_VW = 128;
constexpr int MLP_IN = 195;   // 15 * (1 + 2*6)
constexpr int MLP_IN_PAD = 200;
constexpr int LDX = 204;      // X0 row stride (floats)
constexpr int LDA = 260;      // ACT row stride (floats)
constexpr int L_VOX = 6;

// packed weight offsets (floats)
constexpr int CH0 = MLP_IN_PAD / 8;            // 25 k-chunks
constexpr int CHW = MLP_W / 8;                 // 32
constexpr int CH5 = CH0 + CHW;                 // 57
constexpr int OFF_L0 = 0;
constexpr int OFF_L1 = OFF_L0 + 8 * CH0 * 256;           // 51200
constexpr int OFF_L5 = OFF_L1 + 4 * 8 * CHW * 256;       // 313344
constexpr int OFF_L6 = OFF_L5 + 8 * CH5 * 256;           // 430080
constexpr int OFF_FEAT = OFF_L6 + 2 * 8 * CHW * 256;     // 561152
constexpr int OFF_VIEW = OFF_FEAT + 8 * CHW * 256;       // 626688
constexpr int PACKED_TOTAL = OFF_VIEW + 4 * CHW * 256;   // 659456
static_assert(PACKED_TOTAL == DANBO_MLP_PACKED_FLOATS, "header constant out of date");

struct MlpPackArgs {
    const float* pts_w[8];
    const float* feature_w;
    const float* views_w;
    int Cv;
};

__device__ __forceinline__ int layer_offset(int layer) {
    switch (layer) {
        case 0: return OFF_L0;
        case 1: case 2: case 3: case 4: return OFF_L1 + (layer - 1) * 8 * CHW * 256;
        case 5: return OFF_L5;
        case 6: case 7: return OFF_L6 + (layer - 6) * 8 * CHW * 256;
        case 8: return OFF_FEAT;
        default: return OFF_VIEW;
    }
}

// one thread per packed float
__global__ __launch_bounds__(256) void k_mlp_pack(MlpPackArgs a, float* __restrict__ packed,
                                                  float* __restrict__ views_w_ray_t) {
    const int total = PACKED_TOTAL + a.Cv * MLP_VW;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        if (i >= PACKED_TOTAL) {  // per-ray slice of views_linears.0, transposed to [Cv][128]
            const int q = i - PACKED_TOTAL;
            const int c = q % MLP_VW, k = q / MLP_VW;
            views_w_ray_t[q] = a.views_w[(size_t)c * (MLP_W + a.Cv) + MLP_W + k];
            continue;
        }
        int layer = 9;
        if (i < OFF_L1) layer = 0;
        else if (i < OFF_L5) layer = 1 + (i - OFF_L1) / (8 * CHW * 256);
        else if (i < OFF_L6) layer = 5;
        else if (i < OFF_FEAT) layer = 6 + (i - OFF_L6) / (8 * CHW * 256);
        else if (i < OFF_VIEW) layer = 8;
        const int nch = layer == 0 ? CH0 : (layer == 5 ? CH5 : CHW);
        const int q = i - layer_offset(layer);
        const int t = q & 3, lane = (q >> 2) & 63, kc = (q >> 8) % nch, ct = (q >> 8) / nch;
        const int n = ct * 32 + (lane & 31);
        const int kp = kc * 8 + 4 * (lane >> 5) + t;
        float v = 0.f;
        if (layer == 0) {
            if (kp < MLP_IN) v = a.pts_w[0][(size_t)n * MLP_IN + kp];
        } else if (layer == 5) {
            const int ld = MLP_IN + MLP_W;
            if (kp < MLP_IN) v = a.pts_w[5][(size_t)n * ld + kp];
            else if (kp >= MLP_IN_PAD) v = a.pts_w[5][(size_t)n * ld + MLP_IN + (kp - MLP_IN_PAD)];
        } else if (layer <= 7) {
            v = a.pts_w[layer][(size_t)n * MLP_W + kp];
        } else if (layer == 8) {
            v = a.feature_w[(size_t)n * MLP_W + kp];
        } else {
            v = a.views_w[(size_t)n * (MLP_W + a.Cv) + kp];
        }
        packed[i] = v;
    }
}

// colour head shared by the MLP kernel and the empty-sample path of k_view_consts so that both
// produce bit-identical logits:  sequential fmaf over k = 0..127, bias added last.
__device__ __forceinline__ float rgb_dot(const float* x /*LDS, 16-B aligned*/, const float* __restrict__ w, float b) {
    float acc = 0.f;
#pragma unroll 8
    for (int k = 0; k < MLP_VW; k += 4) {
        const float4 xv = *reinterpret_cast<const float4*>(x + k);
        acc = fmaf(xv.x, w[k], acc);
        acc = fmaf(xv.y, w[k + 1], acc);
        acc = fmaf(xv.z, w[k + 2], acc);
        acc = fmaf(xv.w, w[k + 3], acc);
    }
    return acc + b;
}

// the same head in the summation order of k_pe_mlp16 (csrc/k_mlp16.hip): a sample's 128 features are
// split over four lane groups q (n = 16T + 4q + i), each group accumulates in (T, i) order, then
// ((p0 + p1) + (p2 + p3)) + b  (the xor-16 / xor-32 butterfly).
__device__ __forceinline__ float rgb_dot_halves(const float* x, const float* __restrict__ w, float b) {
    float p[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 4; ++q)
        for (int T = 0; T < 8; ++T) {
            const int n = 16 * T + 4 * q;
            const float4 xv = *reinterpret_cast<const float4*>(x + n);
            p[q] = fmaf(xv.x, w[n], p[q]);
            p[q] = fmaf(xv.y, w[n + 1], p[q]);
            p[q] = fmaf(xv.z, w[n + 2], p[q]);
            p[q] = fmaf(xv.w, w[n + 3], p[q]);
        }
    return ((p[0] + p[1]) + (p[2] + p[3])) + b;
}

// ======================================================================================
// per-ray view constants
// ======================================================================================
// table[c][n] = views_b[n] + sum_k W_view[n][256 + Cpe + k] * code_c[k]   for every frame code c and,
// in row n_codes, for the mean code (Optcodes eval with idx < 0).  One workgroup per code.
__global__ __launch_bounds__(128) void k_view_code_table(const float* __restrict__ framecodes,
                                                         const float* __restrict__ mean_code, int n_codes, int Cf,
                                                         const float* __restrict__ wt_code /*[Cf][128]*/,
                                                         const float* __restrict__ views_b,
                                                         float* __restrict__ table) {
    const int c = blockIdx.x, n = threadIdx.x;
    const float* code = c < n_codes ? framecodes + (size_t)c * Cf : mean_code;
    float acc = 0.f;
    for (int k = 0; k < Cf; ++k) acc = fmaf(code[k], wt_code[(size_t)k * MLP_VW + n], acc);
    table[(size_t)c * MLP_VW + n] = acc + views_b[n];
}

// RPB rays per workgroup iteration.  An iteration is a chain of five barriers with a global-memory round trip behind most of them
// (directions -> pose matrix -> code row): 17.8 us per iteration whatever RPB is -- RPB = 64 (large batches) divides the number of
// iterations by four; RPB = 16 keeps a training-sized batch spread over many workgroups.  The matrix product runs on 8 x 4
// register tiles (see there).  Per-ray arithmetic (summation orders) depends on neither.
template <int RPB>
__global__ __launch_bounds__(256) void k_view_consts(const float* __restrict__ rays_d, const float* __restrict__ skts,
                                                     int R, int G, int ray_mode, int normalise, int L_view,
                                                     const float* __restrict__ framecodes, int n_codes, int Cf,
                                                     const float* __restrict__ mean_code,
                                                     const int64_t* __restrict__ cam_idx,
                                                     const float* __restrict__ wt /*[Cv][128]*/,
                                                     const float* __restrict__ views_b, const float* __restrict__ rgb_w,
                                                     const float* __restrict__ rgb_b,
                                                     const float* __restrict__ empty_consts, int rgb_order,
                                                     const float* __restrict__ code_table,
                                                     const int32_t* __restrict__ ray_list,
                                                     const int32_t* __restrict__ ray_count,
                                                     float* __restrict__ cview, float* __restrict__ raw_empty) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int Cpe = 3 * (1 + 2 * L_view);
    // with a code table the frame-code part (and the bias) is a per-camera constant: only PE(dir) is summed
    const int Cv = code_table ? Cpe : Cpe + Cf;
    if (code_table) Cf = 0;
    float* s_w = smem;                        // [Cv][128]
    float* s_v = s_w + Cv * MLP_VW;           // [Cv][RPB]   (transposed: input-major, ray-minor)
    float* s_x = s_v + Cv * RPB;              // [RPB][128]
    float* s_rgbw = s_x + RPB * MLP_VW;       // [3][128]
    int* s_row = reinterpret_cast<int*>(s_rgbw + 3 * MLP_VW);   // [RPB] code-table row of each ray
    int* s_ray = s_row + RPB;                                   // [RPB] ray of each slot of the iteration (-1: none)
    // ray_list / ray_count: only the listed rays (k_flat_rays' list: the others are rays of constants, nobody reads their rows)
    const int n = ray_list ? min(max(*ray_count, 0), R) : R;
    const int tid = threadIdx.x;
    for (int i = tid; i < Cv * MLP_VW; i += 256) s_w[i] = wt[i];
    for (int i = tid; i < 3 * MLP_VW; i += 256) s_rgbw[i] = rgb_w[i];
    const int rays_per_pose = R / G;

    auto ray_of = [&](int i) { i = min(i, n - 1); return ray_list ? min(max(ray_list[i], 0), R - 1) : i; };
    for (int r0 = blockIdx.x * RPB; r0 < n; r0 += gridDim.x * RPB) {
        __syncthreads();
        for (int rl = tid; rl < RPB; rl += 256) s_ray[rl] = r0 + rl < n ? ray_of(r0 + rl) : -1;
        // ---- build the per-ray view vectors [PE(dir) | frame code], stored [i][ray] ----
        for (int q = tid; q < RPB * 3; q += 256) {
            const int rl = q / 3, k = q % 3;
            const int r = ray_of(r0 + rl);
            float d[3] = {rays_d[3 * r], rays_d[3 * r + 1], rays_d[3 * r + 2]};
            if (ray_mode == 1) {
                const float* M = skts + (size_t)min(r / rays_per_pose, G - 1) * J * 16;  // bone 0 = root
                float t[3];
                for (int a = 0; a < 3; ++a)
                    t[a] = add_rn(add_rn(mul_rn(M[4 * a], d[0]), mul_rn(M[4 * a + 1], d[1])), mul_rn(M[4 * a + 2], d[2]));
                d[0] = t[0]; d[1] = t[1]; d[2] = t[2];
            }
            if (normalise) {
                const float nrm = norm3_torch(d[0], d[1], d[2]);
                const float den = fmaxf(nrm, 1e-12f);
                d[0] = div_rn(d[0], den); d[1] = div_rn(d[1], den); d[2] = div_rn(d[2], den);
            }
            s_v[k * RPB + rl] = d[k];
        }
        if (code_table) {
            for (int rl = tid; rl < RPB; rl += 256) {
                const int r = ray_of(r0 + rl);
                const long idx = cam_idx ? (long)cam_idx[r] : -1;
                s_row[rl] = idx < 0 ? n_codes : (int)min(idx, (long)n_codes - 1);
            }
        }
        for (int i = tid; i < RPB * Cf; i += 256) {
            const int rl = i / Cf, k = i % Cf;
            const int r = ray_of(r0 + rl);
            const long idx = cam_idx ? (long)cam_idx[r] : -1;
            const float val = idx < 0 ? mean_code[k] : framecodes[(size_t)min(idx, (long)n_codes - 1) * Cf + k];
            s_v[(Cpe + k) * RPB + rl] = val;
        }
        __syncthreads();
        // one (ray, axis, level) per thread: sin / cos of 2^l d
        for (int q = tid; q < RPB * 3 * L_view; q += 256) {
            const int rl = q % RPB, k = (q / RPB) % 3, l = q / (RPB * 3);
            float sn, cs;
            pe_sincos(mul_rn(s_v[k * RPB + rl], (float)(1 << l)), &sn, &cs);
            s_v[(3 * (1 + 2 * l) + k) * RPB + rl] = sn;
            s_v[(3 * (2 + 2 * l) + k) * RPB + rl] = cs;
        }
        __syncthreads();
        // ---- cview[r][c] = sum_i W[c][256+i] v[i] + b[c]   (sequential fmaf over i) ----
        // A thread owns a tile of 8 rays x 4 columns: per input i one b128 of weights (columns 4 cg .. 4 cg + 3, conflict-free) and
        // two broadcast b128 of the rays' v_i feed 32 FMAs -- 3 LDS reads per 32 FMAs.  (Round 4's first version, one column and
        // RPB / 2 rays per thread, read 8.25 b128 per 32 FMAs and was bound by exactly those broadcast reads: 97 us for 262 144
        // rays, half of it the LDS pipe.)  Threads 32 rg + cg: column group cg, ray group rg (RPB / 8 groups; RPB = 16: 64 threads).
        const int cg = tid & 31, rg = tid >> 5;
        const bool tile_on = rg < RPB / 8;
        float acc[8][4];
#pragma unroll
        for (int q = 0; q < 8; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[q][e] = 0.f;
        // the code-table rows of this thread's rays: requested now, added after the sum
        float4 trow[8];
        if (code_table && tile_on) {
#pragma unroll
            for (int q = 0; q < 8; ++q) trow[q] = *reinterpret_cast<const float4*>(code_table + (size_t)s_row[rg * 8 + q] * MLP_VW + 4 * cg);
        }
        if (tile_on) {
            const float* vp = s_v + rg * 8;
            const float* wp = s_w + 4 * cg;
#pragma unroll 2
            for (int i = 0; i < Cv; ++i) {
                const float4 w = *reinterpret_cast<const float4*>(wp + i * MLP_VW);
                const float4 va = *reinterpret_cast<const float4*>(vp + i * RPB);
                const float4 vb = *reinterpret_cast<const float4*>(vp + i * RPB + 4);
                const float v8[8] = {va.x, va.y, va.z, va.w, vb.x, vb.y, vb.z, vb.w};
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    acc[q][0] = fmaf(v8[q], w.x, acc[q][0]);
                    acc[q][1] = fmaf(v8[q], w.y, acc[q][1]);
                    acc[q][2] = fmaf(v8[q], w.z, acc[q][2]);
                    acc[q][3] = fmaf(v8[q], w.w, acc[q][3]);
                }
            }
            const float4 b4 = *reinterpret_cast<const float4*>(views_b + 4 * cg);
            const float4 e4 = empty_consts ? *reinterpret_cast<const float4*>(empty_consts + 4 * cg) : float4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int rl = rg * 8 + q, r = s_ray[rl];
                const float4 add = code_table ? trow[q] : b4;
                const float4 a = make_float4(acc[q][0] + add.x, acc[q][1] + add.y, acc[q][2] + add.z, acc[q][3] + add.w);
                if (r >= 0) *reinterpret_cast<float4*>(cview + (size_t)r * MLP_VW + 4 * cg) = a;
                if (empty_consts)
                    *reinterpret_cast<float4*>(s_x + rl * MLP_VW + 4 * cg) =
                        make_float4(fmaxf(e4.x + a.x, 0.f), fmaxf(e4.y + a.y, 0.f), fmaxf(e4.z + a.z, 0.f), fmaxf(e4.w + a.w, 0.f));
            }
        }
        if (empty_consts) {
            __syncthreads();
            if (rgb_order == 2) {
                // the summation order of k_pe_mlp32 (k_mlp32.hip): two lane-group partials per (ray, channel) -- group g walks the
                // features 32 T + 8 Q + 4 g + i in the order T, Q, i -- then (p0 + p1) + b; one partial chain per thread, 16 rays per pass
                for (int pass = 0; pass < RPB / 16; ++pass) {
                    float part = 0.f;
                    const int rl = pass * 16 + (tid >> 4), ch = (tid >> 2) & 3, q = tid & 3;
                    if (ch < 3 && q < 2) {
                        const float* x = s_x + rl * MLP_VW;
                        const float* wv = s_rgbw + ch * MLP_VW;
#pragma unroll
                        for (int TQ = 0; TQ < 16; ++TQ) {
                            const int n = 8 * TQ + 4 * q;
                            const float4 xv = *reinterpret_cast<const float4*>(x + n);
                            const float4 ww = *reinterpret_cast<const float4*>(wv + n);
                            part = fmaf(xv.x, ww.x, part);
                            part = fmaf(xv.y, ww.y, part);
                            part = fmaf(xv.z, ww.z, part);
                            part = fmaf(xv.w, ww.w, part);
                        }
                    }
                    const float p1 = __shfl_xor(part, 1, 64);
                    const int r = s_ray[rl];
                    if (q == 0 && r >= 0) raw_empty[(size_t)r * 4 + ch] = ch < 3 ? (part + p1) + rgb_b[ch] : empty_consts[MLP_VW];
                }
            } else if (rgb_order) {
                // the summation order of k_pe_mlp16: four lane-group partials per (ray, channel), then
                // ((p0 + p1) + (p2 + p3)) + b -- one partial chain per thread, 16 rays per pass
                for (int pass = 0; pass < RPB / 16; ++pass) {
                    float part = 0.f;
                    const int rl = pass * 16 + (tid >> 4), ch = (tid >> 2) & 3, q = tid & 3;
                    if (ch < 3) {
                        const float* x = s_x + rl * MLP_VW;
                        const float* wv = s_rgbw + ch * MLP_VW;
#pragma unroll
                        for (int T = 0; T < 8; ++T) {
                            const int n = 16 * T + 4 * q;
                            const float4 xv = *reinterpret_cast<const float4*>(x + n);
                            const float4 ww = *reinterpret_cast<const float4*>(wv + n);
                            part = fmaf(xv.x, ww.x, part);
                            part = fmaf(xv.y, ww.y, part);
                            part = fmaf(xv.z, ww.z, part);
                            part = fmaf(xv.w, ww.w, part);
                        }
                    }
                    // lanes 4k..4k+3 hold p0..p3 of one (ray, channel)
                    const float p1 = __shfl_xor(part, 1, 64);
                    const float pair = (q & 1) ? p1 + part : part + p1;      // p0+p1 in lanes q=0,1 ; p2+p3 in lanes q=2,3
                    const float other = __shfl_xor(pair, 2, 64);
                    const int r = s_ray[rl];
                    if (q == 0 && r >= 0) raw_empty[(size_t)r * 4 + ch] = ch < 3 ? (pair + other) + rgb_b[ch] : empty_consts[MLP_VW];
                }
            } else {
                for (int q = tid; q < RPB * 4; q += 256) {
                    const int rl = q >> 2, ch = q & 3;
                    const int r = s_ray[rl];
                    if (r >= 0) {
                        float val = empty_consts[MLP_VW];
                        if (ch < 3) val = rgb_dot(s_x + rl * MLP_VW, s_rgbw + ch * MLP_VW, rgb_b[ch]);
                        raw_empty[(size_t)r * 4 + ch] = val;
                    }
                }
            }
        }
    }
}

// ======================================================================================
// the fused MLP
// ======================================================================================
struct MlpArgs {
    const float* h;
    const int32_t* list;
    const int32_t* count;
    int n_cap;
    int S;
    const float* packed;
    const float* pts_b[8];
    const float* alpha_w;
    const float* alpha_b;
    const float* feature_b;
    const float* cview;
    const float* rgb_w;
    const float* rgb_b;
    float* raw_out;
    float* aux_out;
};

// acc[rt][ct] += A[64 x 8*nch] (LDS, row stride lda) * Wp   for this wavefront's NCT column tiles
template <int NCT>
__device__ __forceinline__ void gemm_accumulate(f32x16 (&acc)[2][NCT], const float* __restrict__ A, int lda, int nch,
                                                const float4* __restrict__ wp /* + lane */, int ct_stride /*float4*/) {
    const int lane = threadIdx.x & 63;
    const float* a_ptr0 = A + (lane & 31) * lda + 4 * (lane >> 5);
    const float* a_ptr1 = a_ptr0 + 32 * lda;
    float4 bq[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) bq[ct] = wp[ct * ct_stride];
    float4 a0 = *reinterpret_cast<const float4*>(a_ptr0);
    float4 a1 = *reinterpret_cast<const float4*>(a_ptr1);
#pragma unroll 1
    for (int kc = 0; kc < nch; ++kc) {
        float4 bn[NCT];
        float4 a0n = a0, a1n = a1;
        if (kc + 1 < nch) {
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) bn[ct] = wp[ct * ct_stride + (kc + 1) * 64];
            a0n = *reinterpret_cast<const float4*>(a_ptr0 + (kc + 1) * 8);
            a1n = *reinterpret_cast<const float4*>(a_ptr1 + (kc + 1) * 8);
        } else {
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) bn[ct] = bq[ct];
        }
        const float av0[4] = {a0.x, a0.y, a0.z, a0.w};
        const float av1[4] = {a1.x, a1.y, a1.z, a1.w};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                const float bv = t == 0 ? bq[ct].x : (t == 1 ? bq[ct].y : (t == 2 ? bq[ct].z : bq[ct].w));
                acc[0][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[t], bv, acc[0][ct], 0, 0, 0);
                acc[1][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[t], bv, acc[1][ct], 0, 0, 0);
            }
        }
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) bq[ct] = bn[ct];
        a0 = a0n;
        a1 = a1n;
    }
}

template <int NCT>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[2][NCT]) {
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[rt][ct][i] = 0.f;
}

// ACT[row][col] = act(acc + bias[col]);  C/D layout of the 32x32 MFMA:
// col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
template <int NCT, bool RELU>
__device__ __forceinline__ void store_act(const f32x16 (&acc)[2][NCT], float* __restrict__ ACT,
                                          const float* __restrict__ bias, int col_base) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
        const int col = col_base + ct * 32 + (lane & 31);
        const float b = bias[col];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int row = rt * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
                float v = acc[rt][ct][reg] + b;
                if (RELU) v = fmaxf(v, 0.f);
                ACT[row * LDA + col] = v;
            }
        }
    }
}

constexpr int MLP_LDS_FLOATS = MLP_BM * LDX + MLP_BM * LDA + 4 * MLP_BM /*alpha partials*/ + MLP_BM /*alpha*/ +
                               MLP_BM * 4 /*rgb*/ + MLP_BM /*ray*/ + MLP_BM /*dst*/;

__global__ __launch_bounds__(256, 1) void k_pe_mlp(MlpArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* X0 = smem;
    float* ACT = X0 + MLP_BM * LDX;
    float* s_part = ACT + MLP_BM * LDA;   // [4][64]
    float* s_alpha = s_part + 4 * MLP_BM;  // [64]
    float* s_rgb = s_alpha + MLP_BM;       // [64][4]
    int* s_ray = reinterpret_cast<int*>(s_rgb + MLP_BM * 4);  // [64]
    int* s_dst = s_ray + MLP_BM;                               // [64]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = resolve_count(a.count, a.n_cap);
    const int ntiles = (n + MLP_BM - 1) / MLP_BM;
    const float4* wp = reinterpret_cast<const float4*>(a.packed) + lane;

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int row0 = tile * MLP_BM;
        __syncthreads();  // previous tile completely consumed
        // ---------------- prologue: X0 = PE(h), per-row bookkeeping ----------------
        {
            const int row = tid >> 2, part = tid & 3;
            const int grow = row0 + row;
            float4 hv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (grow < n) hv = reinterpret_cast<const float4*>(a.h + (size_t)grow * DANBO_H_STRIDE)[part];
            const float hvv[4] = {hv.x, hv.y, hv.z, hv.w};
            float* xr = X0 + row * LDX;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = part * 4 + e;
                if (k < FEAT) {
                    const float v = hvv[e];
                    xr[k] = v;
#pragma unroll
                    for (int l = 0; l < L_VOX; ++l) {
                        float sn, cs;
                        sincosf(v * (float)(1 << l), &sn, &cs);
                        xr[FEAT * (1 + 2 * l) + k] = sn;
                        xr[FEAT * (2 + 2 * l) + k] = cs;
                    }
                }
            }
            if (part == 3) {
#pragma unroll
                for (int k = MLP_IN; k < LDX; ++k) xr[k] = 0.f;
            }
            if (part == 0) {
                int m = -1;
                if (grow < n) m = a.list ? a.list[grow] : grow;
                s_dst[row] = m;
                s_ray[row] = m >= 0 ? m / a.S : 0;
            }
        }
        __syncthreads();

        f32x16 acc[2][2];
        // ---------------- density trunk ----------------
#pragma unroll 1
        for (int layer = 0; layer < 8; ++layer) {
            zero_acc<2>(acc);
            // segment 1: X0 for the input layer and for the skip layer, ACT otherwise
            const bool from_x0 = (layer == 0) || (layer == 5);
            const int nch_total = layer == 0 ? CH0 : (layer == 5 ? CH5 : CHW);
            const float4* wl = wp + layer_offset(layer) / 4 + (wave * 2) * nch_total * 64;
            gemm_accumulate<2>(acc, from_x0 ? X0 : ACT, from_x0 ? LDX : LDA, from_x0 ? CH0 : CHW, wl, nch_total * 64);
            if (layer == 5) gemm_accumulate<2>(acc, ACT, LDA, CHW, wl + CH0 * 64, nch_total * 64);
            __syncthreads();  // every wavefront finished reading ACT
            store_act<2, true>(acc, ACT, a.pts_b[layer], wave * 64);
            __syncthreads();
        }
        // ---------------- density logit: alpha = h8 . alpha_w + b ----------------
        {
            const int row = lane, q = wave;
            const float* hr = ACT + row * LDA + q * 64;
            const float* wq = a.alpha_w + q * 64;
            float s = 0.f;
#pragma unroll 4
            for (int k = 0; k < 64; k += 4) {
                const float4 xv = *reinterpret_cast<const float4*>(hr + k);
                s = fmaf(xv.x, wq[k], s);
                s = fmaf(xv.y, wq[k + 1], s);
                s = fmaf(xv.z, wq[k + 2], s);
                s = fmaf(xv.w, wq[k + 3], s);
            }
            s_part[q * MLP_BM + row] = s;
        }
        // ---------------- feature = feature_linear(h8) (no activation) ----------------
        zero_acc<2>(acc);
        gemm_accumulate<2>(acc, ACT, LDA, CHW, wp + OFF_FEAT / 4 + (wave * 2) * CHW * 64, CHW * 64);
        __syncthreads();
        if (tid < MLP_BM)
            s_alpha[tid] = ((s_part[tid] + s_part[MLP_BM + tid]) + s_part[2 * MLP_BM + tid]) + s_part[3 * MLP_BM + tid] + a.alpha_b[0];
        store_act<2, false>(acc, ACT, a.feature_b, wave * 64);
        __syncthreads();
        // ---------------- view layer: relu(W_v[:, :256] feature + cview[ray]) ----------------
        {
            f32x16 accv[2][1];
            zero_acc<1>(accv);
            gemm_accumulate<1>(accv, ACT, LDA, CHW, wp + OFF_VIEW / 4 + wave * CHW * 64, CHW * 64);
            __syncthreads();
            const int col = wave * 32 + (lane & 31);
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int row = rt * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
                    const float pre = accv[rt][0][reg];
                    const float cv = a.cview ? a.cview[(size_t)s_ray[row] * MLP_VW + col] : 0.f;
                    ACT[row * LDA + col] = fmaxf(pre + cv, 0.f);
                    if (a.aux_out && s_dst[row] >= 0) a.aux_out[(size_t)(row0 + row) * (MLP_VW + 1) + col] = pre;
                }
            }
        }
        __syncthreads();
        // ---------------- rgb head + output ----------------
        if (wave < 3) s_rgb[lane * 4 + wave] = rgb_dot(ACT + lane * LDA, a.rgb_w + wave * MLP_VW, a.rgb_b[wave]);
        __syncthreads();
        if (tid < MLP_BM) {
            const int m = s_dst[tid];
            if (m >= 0) {
                const float al = s_alpha[tid];
                reinterpret_cast<float4*>(a.raw_out)[m] = make_float4(s_rgb[tid * 4], s_rgb[tid * 4 + 1], s_rgb[tid * 4 + 2], al);
                if (a.aux_out) a.aux_out[(size_t)(row0 + tid) * (MLP_VW + 1) + MLP_VW] = al;
            }
        }
    }
}

}  // namespace danbo

using namespace danbo;

extern "C" int danbo_mlp_pack(const float* const* pts_w, const float* feature_w, const float* views_w, int Cv,
                               float* packed, float* views_w_ray_t, void* stream) {
    DANBO_CHECK_ARG(pts_w && feature_w && views_w && packed && Cv >= 0 && (Cv == 0 || views_w_ray_t));
    MlpPackArgs a;
    for (int i = 0; i < 8; ++i) a.pts_w[i] = pts_w[i];
    a.feature_w = feature_w;
    a.views_w = views_w;
    a.Cv = Cv;
    hipLaunchKernelGGL(k_mlp_pack, dim3(1024), dim3(256), 0, (hipStream_t)stream, a, packed, views_w_ray_t);
    DANBO_LAUNCH_RET();
}

extern "C" int danbo_view_consts(const float* rays_d, const float* skts, int R, int G, int ray_mode, int normalise,
                                  int L_view, const float* framecodes, int n_codes, int Cf, const float* mean_code,
                                  const int64_t* cam_idx, const float* views_w_ray_t, const float* views_b,
                                  const float* rgb_w, const float* rgb_b, const float* empty_consts, int rgb_order,
                                  const float* code_table, const int32_t* ray_list, const int32_t* ray_count, float* cview,
                                  float* raw_empty, void* stream) {
    DANBO_CHECK_ARG(R > 0 && G > 0 && R % G == 0 && L_view >= 0 && Cf >= 0);
    DANBO_CHECK_ARG((ray_list == nullptr) == (ray_count == nullptr));
    DANBO_CHECK_ARG(Cf == 0 || (mean_code != nullptr && (cam_idx == nullptr || framecodes != nullptr)));
    DANBO_CHECK_ARG((empty_consts == nullptr) == (raw_empty == nullptr));
    // k_view_consts reads these three as float4: a 4-byte aligned slice of a packed parameter blob would fault
    DANBO_CHECK_ARG((uintptr_t)views_b % 16 == 0 && (uintptr_t)empty_consts % 16 == 0 && (uintptr_t)code_table % 16 == 0);
    const int Cv = 3 * (1 + 2 * L_view) + (code_table ? 0 : Cf);
    const int rpb = R >= 16384 ? 64 : 16;      // (32 and 16 rays per iteration on the 512 x 512 frame: 95 and 117 us against 74)
    const size_t lds = sizeof(float) * ((size_t)Cv * MLP_VW + (size_t)rpb * Cv + (size_t)rpb * MLP_VW + 3 * MLP_VW + 2 * rpb);
    DANBO_CHECK_ARG(lds <= 160 * 1024);
    const void* fn = rpb == 64 ? reinterpret_cast<const void*>(k_view_consts<64>) : reinterpret_cast<const void*>(k_view_consts<16>);
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    const int iters = ceil_div(R, rpb);
    const int per_cu = (int)((160 * 1024) / (lds + 1024)) < 8 ? (int)((160 * 1024) / (lds + 1024)) : 8;  // latency-bound phases: fill the CU
    const int grid = iters < num_cu() * per_cu ? iters : num_cu() * (per_cu > 0 ? per_cu : 1);
    if (rpb == 64)
        hipLaunchKernelGGL(k_view_consts<64>, dim3(grid), dim3(256), lds, (hipStream_t)stream, rays_d, skts, R, G, ray_mode,
                           normalise, L_view, framecodes, n_codes, Cf, mean_code, cam_idx, views_w_ray_t, views_b, rgb_w,
                           rgb_b, empty_consts, rgb_order, code_table, ray_list, ray_count, cview, raw_empty);
    else
        hipLaunchKernelGGL(k_view_consts<16>, dim3(grid), dim3(256), lds, (hipStream_t)stream, rays_d, skts, R, G, ray_mode,
                           normalise, L_view, framecodes, n_codes, Cf, mean_code, cam_idx, views_w_ray_t, views_b, rgb_w,
                           rgb_b, empty_consts, rgb_order, code_table, ray_list, ray_count, cview, raw_empty);
    DANBO_LAUNCH_RET();
}

extern "C" int danbo_view_code_table(const float* framecodes, const float* mean_code, int n_codes, int Cf, int L_view,
                                      const float* views_w_ray_t, const float* views_b, float* table, void* stream) {
    DANBO_CHECK_ARG(framecodes && mean_code && n_codes > 0 && Cf > 0 && views_w_ray_t && views_b && table);
    const int Cpe = 3 * (1 + 2 * L_view);
    hipLaunchKernelGGL(k_view_code_table, dim3(n_codes + 1), dim3(128), 0, (hipStream_t)stream, framecodes, mean_code,
                       n_codes, Cf, views_w_ray_t + (size_t)Cpe * MLP_VW, views_b, table);
    DANBO_LAUNCH_RET();
}

extern "C" int danbo_pe_mlp_fwd(const float* h, const int32_t* list, const int32_t* count, int n, int S,
                                 const float* packed, const float* const* pts_b, const float* alpha_w,
                                 const float* alpha_b, const float* feature_b, const float* cview, const float* rgb_w,
                                 const float* rgb_b, float* raw_out, float* aux_out, void* stream) {
    DANBO_CHECK_ARG(n >= 0 && S > 0 && h && packed && pts_b && raw_out);
    if (n == 0) return 0;
    MlpArgs a;
    a.h = h; a.list = list; a.count = count; a.n_cap = n; a.S = S; a.packed = packed;
    for (int i = 0; i < 8; ++i) a.pts_b[i] = pts_b[i];
    a.alpha_w = alpha_w; a.alpha_b = alpha_b; a.feature_b = feature_b; a.cview = cview;
    a.rgb_w = rgb_w; a.rgb_b = rgb_b; a.raw_out = raw_out; a.aux_out = aux_out;
    const size_t lds = sizeof(float) * MLP_LDS_FLOATS;
    DANBO_ENSURE_LDS(k_pe_mlp, lds);
    const int ntiles = ceil_div(n, MLP_BM);
    const int grid = ntiles < num_cu() ? ntiles : num_cu();
    hipLaunchKernelGGL(k_pe_mlp, dim3(grid), dim3(256), lds, (hipStream_t)stream, a);
    DANBO_LAUNCH_RET();
}
