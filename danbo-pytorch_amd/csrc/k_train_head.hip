// The view branch of the training step outside the fused trunk (k_mlp16.hip / k_mlp16_bwd.hip).
// Reference: NeRF.forward_view (core/networks/nerf.py:196-209): views_linears.0 on cat[feature_linear(y7), PE(dir), frame code].
// The trunk evaluates feature_linear and the per-sample columns of views_linears.0 merged (W_fv = W_v[:, :256] W_f) and takes
// the per-RAY columns as a constant of the ray:
//     cview[r]  = W_v[:, 256:] vin[r] + (b_v + W_v[:, :256] b_f)                         k_train_cview        (forward)
// Its adjoint is therefore per ray, and the merged matrix' gradient is pulled back to the two layers by parameter-sized GEMMs:
//     d cview[r] = sum over the rows of ray r of d pre_v[row]                            k_train_ray_grad
//     d W_v[:, 256:] = d cview^T vin ;  csum[cam] = sum over the rays of camera cam      k_train_view_grad
//     d W_v[:, :256] = d W_fv W_f^T + d b_eff b_f^T ;  d W_f = W_v[:, :256]^T d W_fv ;  d b_f = W_v[:, :256]^T d b_eff ;
//     d b_v = d b_eff ;  d codes[cam] = W_v[:, code columns]^T csum[cam]                 k_train_head_chain
// (d W_fv [128,256] and d b_eff [128] = sum over rows of d pre_v come from the weight-gradient kernel k_dw16.)
#include "common.hpp"

namespace danbo {

constexpr int HW = 256, HVW = 128;
typedef float f32x4v __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------------------------
// cview[r, f] = b_eff[f] + sum_k vin[r, k] W_v[f, 256 + k]:  128 threads = features, 8 rays per workgroup
// ------------------------------------------------------------------------------------------------------------------
constexpr int CV_RAYS = 8;
__global__ __launch_bounds__(128) DANBO_NO_PK_F32 void k_train_cview(const float* __restrict__ vin, int ldv, int Cv, const float* __restrict__ views_w,
                                                     const float* __restrict__ b_eff, int R, float* __restrict__ cview) {
    __shared__ float s_v[CV_RAYS][160];
    const int f = threadIdx.x;
    const float* wrow = views_w + (size_t)f * (HW + Cv) + HW;
    for (int r0 = blockIdx.x * CV_RAYS; r0 < R; r0 += gridDim.x * CV_RAYS) {
        __syncthreads();
        for (int i = threadIdx.x; i < CV_RAYS * Cv; i += 128) {
            const int rr = i / Cv, k = i % Cv;
            s_v[rr][k] = r0 + rr < R ? vin[(size_t)(r0 + rr) * ldv + k] : 0.f;
        }
        __syncthreads();
        float acc[CV_RAYS];
#pragma unroll
        for (int rr = 0; rr < CV_RAYS; ++rr) acc[rr] = b_eff[f];
        for (int k = 0; k < Cv; ++k) {
            const float w = wrow[k];
#pragma unroll
            for (int rr = 0; rr < CV_RAYS; ++rr) acc[rr] = fmaf(s_v[rr][k], w, acc[rr]);
            // (The compiler pairs these chains on v_pk_fma_f32, shuffling the LDS values into register pairs with v_mov.  That code
            // produced a wrong sum for one ray in the last 16 lanes of a wavefront about once in 200 training steps WHILE K2 (MFMA)
            // RAN ON THE SAME CUs; kept scalar with `asm volatile("" : "+v"(acc[rr]))` it did not -- 0 of 6 000 replays against 9 of
            // 6 000, tools/stress_replay.py with DANBO_TRAIN_LATE_JOIN=1 -- but takes 71 instead of 18 us.  The training step now
            // orders this kernel in front of K2 (csrc/k_train.hip), where the packed code is bit-stable: 0 of 8 000 replays.)
        }
#pragma unroll
        for (int rr = 0; rr < CV_RAYS; ++rr)
            if (r0 + rr < R) cview[(size_t)(r0 + rr) * HVW + f] = acc[rr];
    }
}

// ------------------------------------------------------------------------------------------------------------------
// d cview[ray(row), f] += d pre_v[row, f] over all rows; d pre_v in fragment order.  128 threads = features; a workgroup owns
// a chunk of consecutive rows (rows of a ray are mostly consecutive: running sum, one atomic per feature when the ray changes)
// ------------------------------------------------------------------------------------------------------------------
constexpr int RG_CHUNK = 64;
__global__ __launch_bounds__(128) void k_train_ray_grad(const float* __restrict__ dpre_v, const int32_t* __restrict__ row_ray,
                                                        const int32_t* __restrict__ cnt, int R, float* __restrict__ d_cview) {
    const int rows = cnt[4];
    const int f = threadIdx.x;
    // element (row i, feature f) of a [rows, 128] fragment-order buffer
    const int foff = ((f >> 5) * 2 + ((f >> 4) & 1)) * 256 + 16 * ((f >> 2) & 3) * 4 + (f & 3);
    for (int base = blockIdx.x * RG_CHUNK; base < rows; base += gridDim.x * RG_CHUNK) {
        float acc = 0.f;
        int cur = -1;
        const int end = min(base + RG_CHUNK, rows);
        for (int i0 = base; i0 < end; i0 += 8) {          // eight rows' loads in one batch (same summation order as row by row)
            int ry[8];
            float dv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = min(i0 + u, end - 1);
                ry[u] = row_ray[i];
                dv[u] = dpre_v[(size_t)(i >> 4) * 2048 + foff + (i & 15) * 4];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (i0 + u >= end) break;
                const int ray = ry[u] < 0 ? 0 : (ry[u] >= R ? R - 1 : ry[u]);
                if (ray != cur) {
                    if (cur >= 0 && acc != 0.f) atomicAdd(d_cview + (size_t)cur * HVW + f, acc);
                    cur = ray;
                    acc = 0.f;
                }
                acc += dv[u];
            }
        }
        if (cur >= 0 && acc != 0.f) atomicAdd(d_cview + (size_t)cur * HVW + f, acc);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// g_views_w[f, 256 + k] += sum_r d cview[r, f] vin[r, k]   (blockIdx.x < k-groups of 8 columns, blockIdx.y = ray slice)
// csum[cam(r), f]      += d cview[r, f]                      (the last blockIdx.x; running sum per camera)
// ------------------------------------------------------------------------------------------------------------------
constexpr int VG_K = 8, VG_SLICES = 32, VG_U = 8;      // VG_U rays in flight per iteration
constexpr int VG_PART_LD = 160;                         // row stride of the per-slice partial sums [VG_SLICES][128][VG_PART_LD]
static_assert(DANBO_TRAIN_VG_PART_FLOATS == VG_SLICES * 128 * VG_PART_LD, "include/danbo_hip.h");
__global__ __launch_bounds__(128) void k_train_view_grad(const float* __restrict__ d_cview, const float* __restrict__ vin, int ldv, int Cv,
                                                         int R, const int64_t* __restrict__ cam_idx, int n_codes, float* __restrict__ g_views_w,
                                                         float* __restrict__ csum, float* __restrict__ part) {
    const int f = threadIdx.x;
    const int kgroups = (Cv + VG_K - 1) / VG_K;
    const int per = (R + gridDim.y - 1) / gridDim.y;
    const int r_begin = blockIdx.y * per, r_end = min(r_begin + per, R);
    if ((int)blockIdx.x < kgroups) {
        const int k0 = blockIdx.x * VG_K;
        // the 8 view inputs of a ray are wave-uniform (scalar loads); VG_U rays in flight per iteration.  (The launch is bound by its
        // final atomics -- 96 ray slices: 139 us, 32: 70 -- so few slices and long, well-fed loops)
        float acc[VG_K];
#pragma unroll
        for (int j = 0; j < VG_K; ++j) acc[j] = 0.f;
        const bool full = k0 + VG_K <= Cv;
        int r = r_begin;
        for (; r + VG_U <= r_end && full; r += VG_U) {
            float dc[VG_U];
            f32x4v v[VG_U][2];
#pragma unroll
            for (int u = 0; u < VG_U; ++u) {
                dc[u] = d_cview[(size_t)(r + u) * HVW + f];
                const f32x4v* vp = reinterpret_cast<const f32x4v*>(vin + (size_t)(r + u) * ldv + k0);
                v[u][0] = vp[0];
                v[u][1] = vp[1];
            }
#pragma unroll
            for (int u = 0; u < VG_U; ++u)
#pragma unroll
                for (int j = 0; j < VG_K; ++j) acc[j] = fmaf(dc[u], v[u][j >> 2][j & 3], acc[j]);
        }
        for (; r < r_end; ++r) {
            const float dc = d_cview[(size_t)r * HVW + f];
            const float* v = vin + (size_t)r * ldv + k0;
#pragma unroll
            for (int j = 0; j < VG_K; ++j) acc[j] = fmaf(dc, k0 + j < Cv ? v[j] : 0.f, acc[j]);
        }
        if (part != nullptr) {
            // per-slice partial sums, added up by k_train_head_chain (16 atomics per gradient entry were what this launch spent its
            // time on -- and, running beside the K2 adjoint, that kernel's)
            float* dst = part + ((size_t)blockIdx.y * HVW + f) * VG_PART_LD + k0;
#pragma unroll
            for (int j = 0; j < VG_K; ++j)
                if (k0 + j < Cv) dst[j] = acc[j];
        } else {
#pragma unroll
            for (int j = 0; j < VG_K; ++j)
                if (k0 + j < Cv && acc[j] != 0.f) atomicAdd(g_views_w + (size_t)f * (HW + Cv) + HW + k0 + j, acc[j]);
        }
    } else if (n_codes > 0) {
        // (eight rays' loads in one batch: ray by ray this was a chain of ~100 dependent round trips, the longest block of the launch)
        float acc = 0.f;
        long cur = -1;
        for (int r = r_begin; r < r_end; r += 8) {
            long ci[8];
            float dv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int rr = min(r + u, r_end - 1);
                ci[u] = cam_idx ? cam_idx[rr] : 0;
                dv[u] = d_cview[(size_t)rr * HVW + f];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (r + u >= r_end) break;
                const long c = ci[u] < 0 ? 0 : (ci[u] >= n_codes ? n_codes - 1 : ci[u]);
                if (c != cur) {
                    if (cur >= 0 && acc != 0.f) atomicAdd(csum + cur * HVW + f, acc);
                    cur = c;
                    acc = 0.f;
                }
                acc += dv[u];
            }
        }
        if (cur >= 0 && acc != 0.f) atomicAdd(csum + cur * HVW + f, acc);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// parameter-sized chain rule of the merged feature / view layer and the frame codes; one thread per output element
// ------------------------------------------------------------------------------------------------------------------
struct HeadChainArgs {
    const float *g_wfv /*[128,256]*/, *g_beff /*[128]*/, *csum /*[n_codes,128]*/;
    const float *feature_w /*[256,256]*/, *feature_b /*[256]*/, *views_w /*[128,256+Cv]*/;
    int Cv, n_codes, code_size, code_col0;       // code_col0: first frame-code column of vin (3 (1 + 2 L_view))
    float *g_feature_w, *g_feature_b, *g_views_w, *g_views_b, *g_codes;
    const float* vg_part;                        // [VG_SLICES][128][VG_PART_LD] partial d W_v[:, 256:] of k_train_view_grad, or nullptr
};

// The two parameter-sized products run as 64 x 64 output tiles through LDS (round 5; one thread per output element with its
// operand row read straight from memory was 30 us on the step's critical path: d W_v's threads walked 256 rows of W_f side by side,
// 64 cache lines per load instruction).  Every output is still ONE fmaf chain over k = 0, 1, 2, ... from the same start value:
// the results are the former kernel's bit for bit.
constexpr int HC_T = 64, HC_K = 64;
constexpr int HC_TILES_VA = (HVW / HC_T) * (HW / HC_T), HC_TILES_F = (HW / HC_T) * (HW / HC_T), HC_REST_BLOCKS = 192;

// C[m, n] (m0.., n0..) = start + sum_k A(m, k) B(n, k);  KMAJOR: A(m, k) = A[k * lda + m], B(n, k) = B[k * ldb + n], else A[m * lda + k], B[n * ldb + k]
// K-steps of 64 through LDS, the next step's operands fetched into registers under the current one's arithmetic (a K-step of 16
// without that prefetch paid one global round trip per step: 16 of them made the kernel as slow as the one it replaced).
template <bool KMAJOR, class Start, class Store>
__device__ __forceinline__ void hc_tile(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb, int K, int m0, int n0,
                                        const Start& start, const Store& store) {
    __shared__ float s_a[HC_K][HC_T + 1], s_b[HC_K][HC_T + 1];
    constexpr int PER = HC_K * HC_T / 256;
    const int t = threadIdx.x, ty = t >> 4, tx = t & 15;
    float acc[4][4], ra[PER], rb[PER];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = start(m0 + 4 * ty + i, n0 + 4 * tx + j);
    auto fetch = [&](int k0) {
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            // consecutive threads read consecutive addresses of the operand
            const int e = t + 256 * u;
            const int kk = KMAJOR ? e / HC_T : e % HC_K, mm = KMAJOR ? e % HC_T : e / HC_K;
            ra[u] = KMAJOR ? A[(size_t)(k0 + kk) * lda + m0 + mm] : A[(size_t)(m0 + mm) * lda + k0 + kk];
            rb[u] = KMAJOR ? B[(size_t)(k0 + kk) * ldb + n0 + mm] : B[(size_t)(n0 + mm) * ldb + k0 + kk];
        }
    };
    fetch(0);
    for (int k0 = 0; k0 < K; k0 += HC_K) {
        __syncthreads();
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int e = t + 256 * u;
            const int kk = KMAJOR ? e / HC_T : e % HC_K, mm = KMAJOR ? e % HC_T : e / HC_K;
            s_a[kk][mm] = ra[u];
            s_b[kk][mm] = rb[u];
        }
        __syncthreads();
        if (k0 + HC_K < K) fetch(k0 + HC_K);
#pragma unroll 8
        for (int kk = 0; kk < HC_K; ++kk) {
            float av[4], bv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) av[i] = s_a[kk][4 * ty + i];
#pragma unroll
            for (int j = 0; j < 4; ++j) bv[j] = s_b[kk][4 * tx + j];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(av[i], bv[j], acc[i][j]);
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) store(m0 + 4 * ty + i, n0 + 4 * tx + j, acc[i][j]);
}

__global__ __launch_bounds__(256) DANBO_NO_PK_F32 void k_train_head_chain(HeadChainArgs a) {
    const int ld = HW + a.Cv;
    int blk = blockIdx.x;
    if (blk < HC_TILES_VA) {          // d W_v[f, c] = d b_eff[f] b_f[c] + sum_j d W_fv[f, j] W_f[c, j]
        const int m0 = (blk / (HW / HC_T)) * HC_T, n0 = (blk % (HW / HC_T)) * HC_T;
        hc_tile<false>(a.g_wfv, HW, a.feature_w, HW, HW, m0, n0, [&](int f, int c) { return a.g_beff[f] * a.feature_b[c]; },
                       [&](int f, int c, float v) { a.g_views_w[(size_t)f * ld + c] = v; });
        return;
    }
    blk -= HC_TILES_VA;
    if (blk < HC_TILES_F) {           // d W_f[c, j] = sum_f W_v[f, c] d W_fv[f, j]
        const int m0 = (blk / (HW / HC_T)) * HC_T, n0 = (blk % (HW / HC_T)) * HC_T;
        hc_tile<true>(a.views_w, ld, a.g_wfv, HW, HVW, m0, n0, [](int, int) { return 0.f; },
                      [&](int c, int j, float v) { a.g_feature_w[(size_t)c * HW + j] = v; });
        return;
    }
    blk -= HC_TILES_F;
    // the vector-sized rest, one thread per output element
    const long n_fb = HW, n_vb = HVW, n_c = (long)a.n_codes * a.code_size;
    const long n_vk = a.vg_part ? (long)HVW * a.Cv : 0;
    const long total = n_fb + n_vb + n_c + n_vk;
    for (long idx = (long)blk * blockDim.x + threadIdx.x; idx < total; idx += (long)HC_REST_BLOCKS * blockDim.x) {
        long i = idx;
        if (i >= total - n_vk) {      // d W_v[f, 256 + k] = sum over the ray slices of k_train_view_grad's partial sums (fixed order)
            i -= total - n_vk;
            const int f = (int)(i / a.Cv), k = (int)(i % a.Cv);
            float acc = 0.f;
#pragma unroll
            for (int sl = 0; sl < VG_SLICES; ++sl) acc += a.vg_part[((size_t)sl * HVW + f) * VG_PART_LD + k];
            a.g_views_w[(size_t)f * ld + HW + k] += acc;
            continue;
        }
        if (i < n_fb) {               // d b_f[c] = sum_f W_v[f, c] d b_eff[f]
            float acc = 0.f;
            for (int f = 0; f < HVW; ++f) acc = fmaf(a.views_w[(size_t)f * ld + i], a.g_beff[f], acc);
            a.g_feature_b[i] = acc;
            continue;
        }
        i -= n_fb;
        if (i < n_vb) { a.g_views_b[i] = a.g_beff[i]; continue; }
        i -= n_vb;
        {                             // d codes[cam, k] = sum_f csum[cam, f] W_v[f, 256 + code_col0 + k]
            const int cam = (int)(i / a.code_size), k = (int)(i % a.code_size);
            float acc = 0.f;
            for (int f = 0; f < HVW; ++f) acc = fmaf(a.csum[(size_t)cam * HVW + f], a.views_w[(size_t)f * ld + HW + a.code_col0 + k], acc);
            a.g_codes[i] = acc;
        }
    }
}

}  // namespace danbo

using namespace danbo;

extern "C" int danbo_train_cview(const float* vin, int ldv, int view_ch, const float* views_w, const float* b_eff, int R, float* cview,
                                 void* stream) {
    DANBO_CHECK_ARG(vin && views_w && b_eff && cview && R > 0 && view_ch >= 0 && view_ch <= 160 && ldv >= view_ch);
    const int blocks = (R + CV_RAYS - 1) / CV_RAYS;
    hipLaunchKernelGGL(k_train_cview, dim3(blocks < num_cu() * 8 ? blocks : num_cu() * 8), dim3(128), 0, (hipStream_t)stream, vin, ldv, view_ch,
                       views_w, b_eff, R, cview);
    DANBO_LAUNCH_RET();
}

extern "C" int danbo_train_view_grads(const float* dpre_v, const int32_t* row_ray, const int32_t* cnt, int rows_cap, int R, const float* vin,
                                      int ldv, int view_ch, const int64_t* cam_idx, int n_codes, float* d_cview /*[R,128] zeroed*/,
                                      float* csum /*[n_codes,128] zeroed*/, float* g_views_w /*accumulated*/,
                                      float* vg_part /*DANBO_TRAIN_VG_PART_FLOATS or NULL*/, void* stream) {
    DANBO_CHECK_ARG(dpre_v && row_ray && cnt && vin && d_cview && g_views_w && rows_cap > 0 && R > 0 && view_ch >= 0 && ldv % 4 == 0);
    DANBO_CHECK_ARG(ldv >= view_ch && (uintptr_t)vin % 16 == 0);
    DANBO_CHECK_ARG(n_codes == 0 || csum);
    DANBO_CHECK_ARG(view_ch <= VG_PART_LD);
    const int chunks = (rows_cap + RG_CHUNK - 1) / RG_CHUNK;
    hipLaunchKernelGGL(k_train_ray_grad, dim3(chunks < num_cu() * 8 ? chunks : num_cu() * 8), dim3(128), 0, (hipStream_t)stream, dpre_v, row_ray,
                       cnt, R, d_cview);
    const int kgroups = (view_ch + VG_K - 1) / VG_K;
    hipLaunchKernelGGL(k_train_view_grad, dim3(kgroups + 1, VG_SLICES), dim3(128), 0, (hipStream_t)stream, d_cview, vin, ldv, view_ch, R, cam_idx,
                       n_codes, g_views_w, csum, vg_part);
    DANBO_LAUNCH_RET();
}

extern "C" int danbo_train_head_chain(const float* g_wfv, const float* g_beff, const float* csum, const float* feature_w,
                                      const float* feature_b, const float* views_w, int view_ch, int n_codes, int code_size, int code_col0,
                                      float* g_feature_w, float* g_feature_b, float* g_views_w, float* g_views_b, float* g_codes,
                                      const float* vg_part, void* stream) {
    DANBO_CHECK_ARG(g_wfv && g_beff && feature_w && feature_b && views_w && g_feature_w && g_feature_b && g_views_w && g_views_b);
    DANBO_CHECK_ARG(view_ch >= 0 && (n_codes == 0 || (csum && g_codes && code_size > 0 && code_col0 >= 0 && code_col0 + code_size <= view_ch)));
    HeadChainArgs a{g_wfv, g_beff, csum, feature_w, feature_b, views_w, view_ch, n_codes, code_size, code_col0,
                    g_feature_w, g_feature_b, g_views_w, g_views_b, g_codes, vg_part};
    hipLaunchKernelGGL(k_train_head_chain, dim3(HC_TILES_VA + HC_TILES_F + HC_REST_BLOCKS), dim3(256), 0, (hipStream_t)stream, a);
    DANBO_LAUNCH_RET();
}
