// Backward of K1b + K2 (factorised gather -> assignment GNN -> masked sigmoid -> blend) for the training step.
// Reference forward: core/networks/gnn_backbone.py:567-629 (MixGNN), :787-828 (sample_from_volume), core/networks/misc.py:331-351,
// core/networks/danbo.py:299-300,406-415; gradients = what loss.backward() (core/trainer.py:563-576) sends to
// prob_linears.*, graph_net.axis_scale and the per-pose volumes (and from there into the pose GNN, k_pose_bwd.hip).  gfx950 only.
//
// Nothing of the forward is stored: the 1 440 B / row part_feat tensor the unfused forward would have to write is
// RECOMPUTED here from the 23 KB pose volumes (L2-resident), in exact fp32.
//
// Work decomposition: a bone's logit only reaches a loss where the bone is valid for the sample (p_j = s(a_j) valid_j, and the
// soft-softmax term multiplies by valid_j too), so the unit of work is a PAIR (row, valid bone j) -- on average 1.6 per
// in-volume row instead of 24.  Pairs are grouped by bone (k_train_bone_lists): a workgroup serves one bone j, keeps the
// weights that bone touches in LDS (layer 0 of j's tree neighbours, its own layers 1 and 2: 14 KB) and their GRADIENTS in
// registers for its whole chunk of pairs -- one flush of atomics per workgroup instead of per pair.
// A wavefront works on two pairs at a time, one per 32-lane half; lane c of a half owns hidden unit c.
//   forward (recompute)    f_k = gather(vol[g,k], x_k) for k in N(j);  y_k = f_k W0_k;  z0 = sum_k A_jk y_k + b0;  a0 = relu z0
//                          z1 = a0 W1_j + b1_j;  a1 = relu z1;  logit = a1 . w2_j + b2_j;  p = 1.002 s(logit) - 0.001
//   upstream               d h [15] (from the MLP through the PE adjoint),  q = sum_j p_j (forward, h[15]),  label (T_i alpha > 0)
//   d logit = (d h . f_j + c_ss (q - label)) 1.002 s (1 - s),   c_ss = 2 coef / (R (S + Sf))
//   ... d w2, d b2, d W1, d b1, d b0, d A_jk, d W0_k, d f_k = A_jk (dz0 W0_k^T) (+ p d h for k = j)
//   d vol[g,k], d axis_scale[k] from d f_k by the adjoint of the interpolation (window detached, mask not differentiable:
//   gnn_backbone.py:804,808)
#include "common.hpp"

namespace danbo {

constexpr int AB_MAXNB = 6;          // bone + at most 5 tree neighbours (SMPL: 5)
constexpr int AB_W = 32;             // hidden width of the assignment net
constexpr int AB_STRIDE = AB_W + 1;  // padded rows: row-wise and column-wise reads are both conflict-free
constexpr int AB_THREADS = 256;
constexpr int AB_PAIRS = 8;          // pairs in flight per workgroup (4 wavefronts x 2 halves)
// pairs per workgroup: amortises the weight staging and the flush.  Chosen on the device between AB_PPW_MIN and AB_PPW so that the
// workgroups that find pairs fit the 2 x 256 resident slots in ONE round (80 k pairs at 128 each were 640 workgroups: a full
// round and a 25 % one)
constexpr int AB_PPW_MIN = 32, AB_PPW = 128;      // AB_PPW: pairs per sub-batch (the LDS row tables); larger chunks take several.  66 KB of LDS, two workgroups per CU

// the overlaid LDS region (bytes) and the flush's accumulator rows (fp64 words) that take it over
constexpr int AB_OFF_W0 = 0;
constexpr int AB_OFF_W1 = AB_OFF_W0 + AB_MAXNB * FEAT * AB_STRIDE * 4;
constexpr int AB_OFF_PDH = (AB_OFF_W1 + AB_W * AB_STRIDE * 4 + 15) & ~15;
constexpr int AB_OFF_VOL0 = AB_OFF_PDH + AB_PPW * 16 * 4;
constexpr int AB_OVER_BYTES = AB_OFF_VOL0 + 2 * AB_MAXNB * VOL * 4;
constexpr int ACC_W1 = 0;                                   // [cc][c]
constexpr int ACC_W0 = ACC_W1 + AB_W * AB_W;                // [q][t][c]
constexpr int ACC_B1 = ACC_W0 + AB_MAXNB * FEAT * AB_W;     // [c]
constexpr int ACC_W2 = ACC_B1 + AB_W;
constexpr int ACC_B0 = ACC_W2 + AB_W;
constexpr int ACC_ADJ = ACC_B0 + AB_W;                      // [q]
constexpr int ACC_SC = ACC_ADJ + 8;                         // [q][3]
constexpr int ACC_B2 = ACC_SC + 3 * AB_MAXNB + 2;
constexpr int ACC_LSS = ACC_B2 + 1;
constexpr int ACC_N = ACC_LSS + 1;
static_assert(ACC_N * 8 <= AB_OVER_BYTES, "the fp64 accumulators must fit the dead tables");

// LDS accumulation goes through fp64: on gfx950 ds_add_f32 takes ~80 ns per wavefront instruction per CU (measured,
// tools/probe/lds_atomic.hip: 40x a ds_add_u32, 22x a ds_add_f64 -- at every contention level), which made the 8 adds per pair of the
// gather adjoint 49 us and the end-of-workgroup reduction 60 us of this kernel's 274
__device__ __forceinline__ void ab_lds_add(double* p, float v) { atomicAdd(p, (double)v); }

struct ABArgs {
    // geometry
    const float *rays_o, *rays_d, *z_c, *z_f, *skts, *align, *axis_scale, *volumes;
    int R, S, Sf, G;
    // rows
    const int32_t *row_sample, *row_ray, *cnt, *lists, *cntb;
    int cap;
    const float *h_rows, *d_h;
    const uint8_t *label_c, *label_f;
    const uint32_t *bits_c, *bits_f;
    // assignment net (reference parameter layouts)
    const float *w0 /*[24,15,32]*/, *adj_w /*[24,24]*/, *adj /*[24,24]*/, *b0 /*[32]*/, *w1 /*[24,32,32]*/, *b1 /*[24,32]*/, *w2 /*[24,32]*/,
        *b2 /*[24]*/;
    // gradients (accumulated with atomics: the caller zeroes them)
    float *g_w0, *g_adj_w, *g_b0, *g_w1, *g_b1, *g_w2, *g_b2, *g_vol /*[G,24,240]*/, *g_scale /*[24,3]*/;
    float c_ss;          // 2 coef / (R (S + Sf))
    float* loss;         // loss[2] += (label - q)^2 of in-volume rows
    const float* d_p;    // [rows, 24] upstream gradient of the masked probabilities p_j valid_j, or nullptr (torch.ops.danbo.assign_blend)
};

// sum over the 32 lanes of a wave half, in every lane of the half: DPP partial sums inside the 16-lane rows, row 0 / 2's total
// carried into row 1 / 3 (row_bcast15), then lanes 31 and 63 read out -- about ten VALU instructions instead of five dependent
// LDS-crossbar shuffles (the kernel takes ~7 such sums per iteration, each a chain of ~5 x 100 cycles before)
__device__ __forceinline__ float half_sum32(float v) {
    float x = v;
    DANBO_DPP_STEP(dpp_add_, 0.f, 0x111, 0xf) DANBO_DPP_STEP(dpp_add_, 0.f, 0x112, 0xf)
    DANBO_DPP_STEP(dpp_add_, 0.f, 0x114, 0xf) DANBO_DPP_STEP(dpp_add_, 0.f, 0x118, 0xf)
    DANBO_DPP_STEP(dpp_add_, 0.f, 0x142, 0xa)
    const float lo = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 31));
    const float hi = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 63));
    return (threadIdx.x & 32) ? hi : lo;
}

// The per-pair scratch (s_f, s_a0, s_dz1, s_dz0, s_df, s_dh) is private to a 32-lane half, so the layers of one pair only have
// to be ordered inside the wavefront: LDS executes a wavefront's operations in issue order, which leaves a compiler fence --
// no s_barrier (with workgroup barriers the four wavefronts ran in lock step through five latency chains per iteration).
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__global__ __launch_bounds__(AB_THREADS, 2) void k_assign_bwd(ABArgs a, int target_wgs) {
    // s_w0, s_w1, s_pdh and s_vol0 are carved from one region: they are dead once the pair loop is over, and the flush re-uses the
    // region as its fp64 accumulators (AbAcc)
    __shared__ __attribute__((aligned(16))) char s_over[AB_OVER_BYTES];
    auto& s_w0 = *reinterpret_cast<float(*)[AB_MAXNB][FEAT][AB_STRIDE]>(s_over + AB_OFF_W0);
    auto& s_w1 = *reinterpret_cast<float(*)[AB_W][AB_STRIDE]>(s_over + AB_OFF_W1);
    auto& s_pdh = *reinterpret_cast<float(*)[AB_PPW][16]>(s_over + AB_OFF_PDH);
    auto& s_vol0 = *reinterpret_cast<float(*)[2][AB_MAXNB][VOL]>(s_over + AB_OFF_VOL0);
    __shared__ float s_b0[AB_W], s_b1[AB_W], s_w2[AB_W];
    __shared__ float s_adj[AB_MAXNB], s_align[AB_MAXNB][12], s_scale[AB_MAXNB][4];
    __shared__ int s_nb[AB_MAXNB];
    __shared__ int s_nq;
    __shared__ double s_gvol[2][AB_MAXNB][VOL];        // fp64: see ab_lds_add
    // per-pair scratch (one per wave half)
    __shared__ __attribute__((aligned(16))) float s_f[AB_PAIRS][AB_MAXNB][16];
    __shared__ __attribute__((aligned(16))) float s_df[AB_PAIRS][AB_MAXNB][16];
    __shared__ __attribute__((aligned(16))) float s_a0[AB_PAIRS][AB_W], s_dz1[AB_PAIRS][AB_W], s_dz0[AB_PAIRS][AB_W], s_dh[AB_PAIRS][16];

    // workgroup -> (bone j, chunk of its pair list): bones get ceil(pairs / ppw) workgroups each, in bone order
    // (the 24 counters and the bone's adjacency row are fetched by 24 lanes at once: a serial loop of dependent global loads
    //  cost ~50 us per workgroup)
    __shared__ int s_cnt[J];
    __shared__ float s_adjrow[J];
    const int tid = threadIdx.x, lane = tid & 63, c = lane & 31, slot = tid >> 5;   // slot: which pair of the 8 in flight
    if (tid < J) s_cnt[tid] = min(a.cntb[tid], a.cap);
    __syncthreads();
    int total_pairs = 0;
    for (int k = 0; k < J; ++k) total_pairs += s_cnt[k];
    // pairs per workgroup: 128 .. 256 so that the workgroups with pairs fill the resident slots in one round -- and MORE than
    // 256 (processed in sub-batches of 256 below) when the launch grid could not cover the chunks otherwise (heavily overlapping
    // volumes after axis_scale has grown, very large batches): sum_j ceil(cnt_j / ppw) <= total / ppw + J <= gridDim.x, no pair
    // is ever dropped
    const int wgs_avail = max(min(target_wgs, (int)gridDim.x - J), 1);
    const int ppw = max(((total_pairs + wgs_avail - 1) / wgs_avail + 7) & ~7, AB_PPW_MIN);
    int j = 0, wg = blockIdx.x, npairs = 0;
    for (; j < J; ++j) {
        npairs = s_cnt[j];
        const int need = (npairs + ppw - 1) / ppw;
        if (wg < need) break;
        wg -= need;
    }
    if (j == J) return;
    const int p_begin = wg * ppw, p_end = min(p_begin + ppw, npairs);

    // ---- neighbourhood of bone j from the adjacency buffer (self first) and its weights
    if (tid < J) s_adjrow[tid] = a.adj[j * J + tid];
    __syncthreads();
    if (tid == 0) {
        int nq = 0;
        s_nb[nq++] = j;
        for (int k = 0; k < J && nq < AB_MAXNB; ++k)
            if (k != j && s_adjrow[k] != 0.f) s_nb[nq++] = k;
        s_nq = nq;
    }
    __syncthreads();
    const int nq = s_nq;
    for (int i = tid; i < nq * FEAT * AB_W; i += AB_THREADS) {
        const int q = i / (FEAT * AB_W), t = (i / AB_W) % FEAT, cc = i % AB_W;
        s_w0[q][t][cc] = a.w0[((size_t)s_nb[q] * FEAT + t) * AB_W + cc];
    }
    for (int i = tid; i < AB_W * AB_W; i += AB_THREADS) s_w1[i / AB_W][i % AB_W] = a.w1[(size_t)j * AB_W * AB_W + i];
    if (tid < AB_W) { s_b0[tid] = a.b0[tid]; s_b1[tid] = a.b1[j * AB_W + tid]; s_w2[tid] = a.w2[j * AB_W + tid]; }
    if (tid < nq) s_adj[tid] = a.adj_w[j * J + s_nb[tid]] * a.adj[j * J + s_nb[tid]];
    for (int i = tid; i < nq * 12; i += AB_THREADS) s_align[i / 12][i % 12] = a.align[s_nb[i / 12] * 16 + i % 12];
    for (int i = tid; i < nq * 4; i += AB_THREADS) s_scale[i / 4][i % 4] = (i % 4) < 3 ? a.axis_scale[s_nb[i / 4] * 3 + i % 4] : 1.f;
    for (int i = tid; i < 2 * AB_MAXNB * VOL; i += AB_THREADS) (&s_gvol[0][0][0])[i] = 0.0;
    __syncthreads();
    const float b2 = a.b2[j];
    const int first_f = a.cnt[2];
    const int rays_per_pose = a.R / a.G;
    // pose whose volume gradient is gathered in LDS: the pose of this chunk's first pair (rows are ray-ordered, so most pairs
    // of a chunk share it); pairs of other poses go to global memory directly
    // Everything a pair needs from the row tables is fetched ONCE per workgroup, one pair per thread, into LDS: as a chain of
    // three dependent global loads at the top of every iteration it cost ~5 us per iteration with nothing to hide it behind.
    __shared__ int s_pi[AB_PPW], s_pm[AB_PPW], s_pray[AB_PPW];
    __shared__ float s_plab[AB_PPW], s_pq[AB_PPW];
    __shared__ uint32_t s_pbits[AB_PPW];
    __shared__ float s_pp[AB_PPW][3];             // the pair's sample point o + d z
    // two poses' transforms, volumes and volume-gradient accumulators in LDS: g0 (the pose of the chunk's first pair) and g0 + 1 --
    // rows are ray-ordered, a chunk of ~170 pairs of one bone straddles a pose boundary about every other time
    __shared__ float s_skt0[2][AB_MAXNB][12];
    int g0 = 0;
    // ---- gradient accumulators of this lane (hidden unit c of half `slot & 1`)
    // The two halves of a wavefront SHARE one set of weight-gradient accumulators: half h keeps rows cc = 16 h .. 16 h + 15 of
    // d W1 and features t = 8 h .. 8 h + 7 of d W0 -- for BOTH pairs of the wavefront (the other pair's dz1 / dz0 come from its
    // scratch).  Same FMA count as one full set per half, 70 accumulators instead of 134: nothing spills inside the pair loop
    // (a spilled value costs a waited scratch round trip, ~0.5 us, and there were nine per iteration).
    const int h = (lane >> 5) & 1;
    float gw1[AB_W / 2];             // d W1[j][16 h + i][c]
    float gw0[AB_MAXNB][8];          // d W0[nb q][8 h + i][c]   (t = 15: the zero pad of s_f, never flushed)
    float gb1 = 0.f, gw2 = 0.f, gb0 = 0.f, gb2 = 0.f, gadj[AB_MAXNB], gsc = 0.f, lss = 0.f;
#pragma unroll
    for (int i = 0; i < AB_W / 2; ++i) gw1[i] = 0.f;
#pragma unroll
    for (int q = 0; q < AB_MAXNB; ++q) {
        gadj[q] = 0.f;
#pragma unroll
        for (int t = 0; t < 8; ++t) gw0[q][t] = 0.f;
    }
    // lanes 0 .. 3 nq - 1 of a half: (neighbour gq, axis gk) of the gather and of its adjoint
    const int gq = c / 3, gk = c % 3;
    const bool glane = c < 3 * nq;
#pragma unroll 1
    for (int sb_begin = p_begin; sb_begin < p_end; sb_begin += AB_PPW) {      // one pass unless ppw > 256
    const int sb_end = min(sb_begin + AB_PPW, p_end);
    if (sb_begin != p_begin) __syncthreads();                                 // the previous sub-batch's tables are still being read
    if (tid < sb_end - sb_begin) {
        const int i = a.lists[(size_t)j * a.cap + sb_begin + tid];
        const bool coarse = i < first_f;
        const int m = a.row_sample[i];
        const int ray_ = a.row_ray[i];
        s_pi[tid] = i; s_pm[tid] = m; s_pray[tid] = ray_;
        const float z_ = coarse ? a.z_c[m] : a.z_f[m];
        {
            const float o[3] = {a.rays_o[3 * ray_], a.rays_o[3 * ray_ + 1], a.rays_o[3 * ray_ + 2]};
            const float d[3] = {a.rays_d[3 * ray_], a.rays_d[3 * ray_ + 1], a.rays_d[3 * ray_ + 2]};
            float p[3];
            sample_point(o, d, z_, p);
            s_pp[tid][0] = p[0]; s_pp[tid][1] = p[1]; s_pp[tid][2] = p[2];
        }
        s_plab[tid] = (float)(coarse ? a.label_c[m] : a.label_f[m]);
        s_pbits[tid] = coarse ? a.bits_c[m] : a.bits_f[m];
        s_pq[tid] = a.h_rows[(size_t)i * 16 + 15];
        const float4* dh = reinterpret_cast<const float4*>(a.d_h + (size_t)i * 16);
#pragma unroll
        for (int e = 0; e < 4; ++e) reinterpret_cast<float4*>(s_pdh[tid])[e] = dh[e];
    }
    __syncthreads();
    if (sb_begin == p_begin) {
        g0 = min(s_pray[0] / rays_per_pose, a.G - 1);
        // the bone transforms and volumes of the neighbourhood for poses g0 and g0 + 1: LDS instead of two more dependent global
        // round trips per pair
        for (int i = tid; i < 2 * nq * 12; i += AB_THREADS) {
            const int u = i / (nq * 12), r = i % (nq * 12), gg = min(g0 + u, a.G - 1);
            s_skt0[u][r / 12][r % 12] = a.skts[((size_t)gg * J + s_nb[r / 12]) * 16 + r % 12];
        }
        for (int i = tid; i < 2 * nq * VOL; i += AB_THREADS) {
            const int u = i / (nq * VOL), r = i % (nq * VOL), gg = min(g0 + u, a.G - 1);
            s_vol0[u][r / VOL][r % VOL] = a.volumes[((size_t)gg * J + s_nb[r / VOL]) * VOL + r % VOL];
        }
        __syncthreads();
    }

    const int iters = (sb_end - sb_begin + AB_PAIRS - 1) / AB_PAIRS;
    for (int it = 0; it < iters; ++it) {
        const int pl_ = it * AB_PAIRS + slot;
        const bool live = sb_begin + pl_ < sb_end;          // uniform per half
        const int pl = live ? pl_ : 0;
        const int ray = s_pray[pl];
        const int g = min(ray / rays_per_pose, a.G - 1);
        const int gl = g - g0;                              // 0 / 1: this pair's pose is one of the two held in LDS
        const bool in_lds = gl == 0 || (gl == 1 && g0 + 1 < a.G);
        const float lab = s_plab[pl], qrow = s_pq[pl];
        const uint32_t bits = s_pbits[pl];
        if (c < 16) s_dh[slot][c] = c < FEAT ? s_pdh[pl][c] : 0.f;

        // ---- recompute the gather: lane (gq, gk) evaluates neighbour gq completely and keeps axis gk
        float x_k = 0.f, win = 0.f, w0t = 0.f, w1t = 0.f, sck = 1.f;
        int y0 = 0, y1 = 0;
        bool ok0 = false, ok1 = false;
        int o0 = 0, o1 = 0;
        if (glane) {
            const int k = s_nb[gq];
            const float p[3] = {s_pp[pl][0], s_pp[pl][1], s_pp[pl][2]};
            float pl_[3], pt[3], x[3], skt[12];
            // (LDS and global memory take separate branches: a pointer that may be either compiles to FLAT loads, and the
            //  conditional voxel reads to one waited round trip each -- ~17 serial vector-memory round trips per iteration)
            if (in_lds) {
#pragma unroll
                for (int e = 0; e < 12; ++e) skt[e] = s_skt0[gl][gq][e];
            } else {
                const float* src = a.skts + ((size_t)g * J + k) * 16;
#pragma unroll
                for (int e = 0; e < 12; ++e) skt[e] = src[e];
            }
            affine_unfused(skt, p, pl_);
            // (the intermediate point goes through an empty asm: the compiler had paired products of the two transforms on ONE
            // v_pk_mul_f32 with op_sel:[0,1] -- the operand selection that is wrong on gfx950 beside MFMA wavefronts, common.hpp
            // DANBO_NO_PK_F32; compiling the whole kernel without packed instructions instead costs it 57 us, 234 -> 291)
#pragma unroll
            for (int e = 0; e < 3; ++e) asm volatile("" : "+v"(pl_[e]));
            affine_unfused(s_align[gq], pl_, pt);
#pragma unroll
            for (int e = 0; e < 3; ++e) x[e] = div_rn(pt[e], fabsf(s_scale[gq][e]));
            win = coord_window(x);
            x_k = x[gk];
            sck = s_scale[gq][gk];
            const float iy = div_rn(sub_rn(mul_rn(add_rn(x_k, 1.0f), (float)VRES), 1.0f), 2.0f);
            const float fl = floorf(iy);
            w1t = sub_rn(iy, fl);
            w0t = sub_rn(1.0f, w1t);
            y0 = (int)fminf(fmaxf(fl, -2.0f), (float)VRES + 1.0f);
            y1 = y0 + 1;
            ok0 = y0 >= 0 && y0 < VRES;
            ok1 = y1 >= 0 && y1 < VRES;
            o0 = min(max(y0, 0), VRES - 1) * 3 + gk;       // clamped: every voxel read is unconditional, one batch
            o1 = min(max(y1, 0), VRES - 1) * 3 + gk;
            float v0[VOXF], v1[VOXF];
            if (in_lds) {
#pragma unroll
                for (int f = 0; f < VOXF; ++f) { v0[f] = s_vol0[gl][gq][f * (VRES * 3) + o0]; v1[f] = s_vol0[gl][gq][f * (VRES * 3) + o1]; }
            } else {
                const float* vol = a.volumes + ((size_t)g * J + k) * VOL;
#pragma unroll
                for (int f = 0; f < VOXF; ++f) { v0[f] = vol[f * (VRES * 3) + o0]; v1[f] = vol[f * (VRES * 3) + o1]; }
            }
#pragma unroll
            for (int f = 0; f < VOXF; ++f)
                s_f[slot][gq][f * 3 + gk] = mul_rn(add_rn(mul_rn(ok0 ? v0[f] : 0.f, w0t), mul_rn(ok1 ? v1[f] : 0.f, w1t)), win);
            if (gk == 0) s_f[slot][gq][15] = 0.f;
        }
        wave_sync();

        // ---- layer 0: y_q[c], z0[c]
        float y[AB_MAXNB];
        float z0 = s_b0[c];
#pragma unroll
        for (int q = 0; q < AB_MAXNB; ++q) {
            y[q] = 0.f;
            if (q < nq) {
#pragma unroll
                for (int t = 0; t < FEAT; ++t) y[q] = fmaf(s_f[slot][q][t], s_w0[q][t][c], y[q]);
                z0 = fmaf(s_adj[q], y[q], z0);
            }
        }
        const float a0 = fmaxf(z0, 0.f);
        s_a0[slot][c] = a0;
        wave_sync();
        // ---- layer 1, 2
        float z1 = s_b1[c];
#pragma unroll
        for (int cc = 0; cc < AB_W; ++cc) z1 = fmaf(s_a0[slot][cc], s_w1[cc][c], z1);
        const float a1 = fmaxf(z1, 0.f);
        const float logit = half_sum32(a1 * s_w2[c]) + b2;
        const float sg = sigmoidf_(logit);
        const float pj = sg * 1.002f - 0.001f;
        // ---- upstream
        float dp = c < FEAT ? s_dh[slot][c] * s_f[slot][0][c] : 0.f;   // neighbour 0 is the bone itself
        dp = half_sum32(dp);
        if (a.d_p != nullptr) dp += a.d_p[(size_t)s_pi[pl] * J + j];
        float dlogit = (dp + a.c_ss * (qrow - lab)) * 1.002f * sg * (1.0f - sg);
        if (!live) dlogit = 0.f;
        if (live && c == 0 && j == __builtin_ctz(bits)) lss += (lab - qrow) * (lab - qrow);   // once per row
        // ---- layer 2, 1 adjoints
        gw2 = fmaf(dlogit, a1, gw2);
        gb2 += dlogit;
        const float dz1 = z1 > 0.f ? dlogit * s_w2[c] : 0.f;
        gb1 += dz1;
        s_dz1[slot][c] = dz1;
        wave_sync();
        {
            const float dz1_o = s_dz1[slot ^ 1][c];            // the wavefront's other pair
#pragma unroll
            for (int i = 0; i < AB_W / 2; ++i)
                gw1[i] = fmaf(s_a0[slot ^ 1][16 * h + i], dz1_o, fmaf(s_a0[slot][16 * h + i], dz1, gw1[i]));
        }
        float da0 = 0.f;
#pragma unroll
        for (int cc = 0; cc < AB_W; ++cc) da0 = fmaf(s_w1[c][cc], s_dz1[slot][cc], da0);
        const float dz0 = z0 > 0.f ? da0 : 0.f;
        gb0 += dz0;
        s_dz0[slot][c] = dz0;
        wave_sync();
        // ---- layer 0 adjoints: adjacency, W0
        const float dz0_o = s_dz0[slot ^ 1][c];
#pragma unroll
        for (int q = 0; q < AB_MAXNB; ++q) {
            if (q < nq) {
                gadj[q] += half_sum32(dz0 * y[q]);
                const float dy = s_adj[q] * dz0, dy_o = s_adj[q] * dz0_o;
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    gw0[q][i] = fmaf(s_f[slot ^ 1][q][8 * h + i], dy_o, fmaf(s_f[slot][q][8 * h + i], dy, gw0[q][i]));
            }
        }
        // ---- d f_q[t] = A_jq sum_cc dz0[cc] W0_q[t][cc]  (+ p d h[t] for the bone itself): (q, t) pairs spread over the half
        for (int e = c; e < nq * FEAT; e += 32) {
            const int q = e / FEAT, t = e % FEAT;
            float acc = 0.f;
#pragma unroll
            for (int cc = 0; cc < AB_W; ++cc) acc = fmaf(s_dz0[slot][cc], s_w0[q][t][cc], acc);
            acc *= s_adj[q];
            if (q == 0) acc = fmaf(pj, s_dh[slot][t], acc);
            s_df[slot][q][t] = live ? acc : 0.f;
        }
        wave_sync();
        // ---- adjoint of the gather (k_backward.hip's arithmetic): lane (gq, gk)
        if (glane && live && win != 0.f) {
            const int k = s_nb[gq];
            float dx = 0.f, gqv[VOXF];
#pragma unroll
            for (int f = 0; f < VOXF; ++f) gqv[f] = s_df[slot][gq][f * 3 + gk] * win;
            if (in_lds) {
                double* gv = &s_gvol[gl][gq][0];
#pragma unroll
                for (int f = 0; f < VOXF; ++f) {
                    const float v0 = ok0 ? s_vol0[gl][gq][f * (VRES * 3) + o0] : 0.f, v1 = ok1 ? s_vol0[gl][gq][f * (VRES * 3) + o1] : 0.f;
                    if (ok0) ab_lds_add(gv + f * (VRES * 3) + o0, gqv[f] * w0t);
                    if (ok1) ab_lds_add(gv + f * (VRES * 3) + o1, gqv[f] * w1t);
                    dx += gqv[f] * (v1 - v0);
                }
            } else {
                const float* vol = a.volumes + ((size_t)g * J + k) * VOL;
                float* gv = a.g_vol + ((size_t)g * J + k) * VOL;
#pragma unroll
                for (int f = 0; f < VOXF; ++f) {
                    const float v0 = ok0 ? vol[f * (VRES * 3) + o0] : 0.f, v1 = ok1 ? vol[f * (VRES * 3) + o1] : 0.f;
                    if (ok0) atomicAdd(gv + f * (VRES * 3) + o0, gqv[f] * w0t);
                    if (ok1) atomicAdd(gv + f * (VRES * 3) + o1, gqv[f] * w1t);
                    dx += gqv[f] * (v1 - v0);
                }
            }
            dx *= 0.5f * (float)VRES;
            gsc += dx * (-x_k / fabsf(sck)) * (sck < 0.f ? -1.f : 1.f);
        }
        wave_sync();                    // the next iteration's scratch writes come after these reads
    }
    }   // sub-batches

    // ---- flush: the 8 pair slots of the workgroup first add up in LDS (the tables of the overlaid region are dead by now and
    // become fp64 accumulators), then every entry goes to global memory with one atomic
    __syncthreads();
    double* acc = reinterpret_cast<double*>(s_over);
    for (int i = tid; i < ACC_N; i += AB_THREADS) acc[i] = 0.0;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < AB_W / 2; ++i) ab_lds_add(acc + ACC_W1 + (16 * h + i) * AB_W + c, gw1[i]);
#pragma unroll
    for (int q = 0; q < AB_MAXNB; ++q) {
        if (q < nq) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (8 * h + i < FEAT) ab_lds_add(acc + ACC_W0 + (q * FEAT + 8 * h + i) * AB_W + c, gw0[q][i]);
            if (c == 0) ab_lds_add(acc + ACC_ADJ + q, gadj[q]);
        }
    }
    ab_lds_add(acc + ACC_B1 + c, gb1);
    ab_lds_add(acc + ACC_W2 + c, gw2);
    ab_lds_add(acc + ACC_B0 + c, gb0);
    if (c == 0) { ab_lds_add(acc + ACC_B2, gb2); ab_lds_add(acc + ACC_LSS, lss); }
    if (glane) ab_lds_add(acc + ACC_SC + gq * 3 + gk, gsc);
    __syncthreads();
    float* gw1p = a.g_w1 + (size_t)j * AB_W * AB_W;
    for (int i = tid; i < AB_W * AB_W; i += AB_THREADS) {
        const float v = (float)acc[ACC_W1 + i];
        if (v != 0.f) atomicAdd(gw1p + i, v);
    }
    for (int i = tid; i < nq * FEAT * AB_W; i += AB_THREADS) {
        const int q = i / (FEAT * AB_W), r = i % (FEAT * AB_W);
        const float v = (float)acc[ACC_W0 + i];
        if (v != 0.f) atomicAdd(a.g_w0 + (size_t)s_nb[q] * FEAT * AB_W + r, v);
    }
    if (tid < AB_W) {
        const float vb1 = (float)acc[ACC_B1 + tid], vw2 = (float)acc[ACC_W2 + tid], vb0 = (float)acc[ACC_B0 + tid];
        if (vb1 != 0.f) atomicAdd(a.g_b1 + j * AB_W + tid, vb1);
        if (vw2 != 0.f) atomicAdd(a.g_w2 + j * AB_W + tid, vw2);
        if (vb0 != 0.f) atomicAdd(a.g_b0 + tid, vb0);
    }
    if (tid < nq) {
        const float v = (float)acc[ACC_ADJ + tid];
        if (v != 0.f) atomicAdd(a.g_adj_w + j * J + s_nb[tid], v * a.adj[j * J + s_nb[tid]]);
    }
    if (tid < nq * 3) {
        const float v = (float)acc[ACC_SC + tid];
        if (v != 0.f) atomicAdd(a.g_scale + s_nb[tid / 3] * 3 + tid % 3, v);
    }
    if (tid == 0) {
        const float vb2 = (float)acc[ACC_B2], vl = (float)acc[ACC_LSS];
        if (vb2 != 0.f) atomicAdd(a.g_b2 + j, vb2);
        if (vl != 0.f) atomicAdd(a.loss + 2, vl);
    }
    for (int i = tid; i < 2 * nq * VOL; i += AB_THREADS) {
        const int u = i / (nq * VOL), r = i % (nq * VOL);
        const float v = (float)s_gvol[u][r / VOL][r % VOL];
        if (v != 0.f && g0 + u < a.G) atomicAdd(a.g_vol + ((size_t)(g0 + u) * J + s_nb[r / VOL]) * VOL + r % VOL, v);
    }
}

}  // namespace danbo

using namespace danbo;

extern "C" int danbo_assign_blend_bwd(const DanboAssignBwd* p, void* stream) {
    DANBO_CHECK_ARG(p && p->rays_o && p->rays_d && p->z_c && p->z_f && p->skts && p->align && p->axis_scale && p->volumes);
    DANBO_CHECK_ARG(p->R > 0 && p->S > 0 && p->Sf > 0 && p->G > 0 && p->R % p->G == 0 && p->rows_cap > 0);
    DANBO_CHECK_ARG(p->row_sample && p->row_ray && p->cnt && p->lists && p->cntb && p->h_rows && p->d_h && p->label_c && p->label_f);
    DANBO_CHECK_ARG(p->bits_c && p->bits_f && p->w0 && p->adj_w && p->adj && p->b0 && p->w1 && p->b1 && p->w2 && p->b2);
    DANBO_CHECK_ARG(p->g_w0 && p->g_adj_w && p->g_b0 && p->g_w1 && p->g_b1 && p->g_w2 && p->g_b2 && p->g_vol && p->g_scale && p->loss);
    ABArgs a;
    a.rays_o = p->rays_o; a.rays_d = p->rays_d; a.z_c = p->z_c; a.z_f = p->z_f; a.skts = p->skts; a.align = p->align;
    a.axis_scale = p->axis_scale; a.volumes = p->volumes; a.R = p->R; a.S = p->S; a.Sf = p->Sf; a.G = p->G;
    a.row_sample = p->row_sample; a.row_ray = p->row_ray; a.cnt = p->cnt; a.lists = p->lists; a.cntb = p->cntb; a.cap = p->rows_cap;
    a.h_rows = p->h_rows; a.d_h = p->d_h; a.label_c = p->label_c; a.label_f = p->label_f; a.bits_c = p->bits_c; a.bits_f = p->bits_f;
    a.w0 = p->w0; a.adj_w = p->adj_w; a.adj = p->adj; a.b0 = p->b0; a.w1 = p->w1; a.b1 = p->b1; a.w2 = p->w2; a.b2 = p->b2;
    a.g_w0 = p->g_w0; a.g_adj_w = p->g_adj_w; a.g_b0 = p->g_b0; a.g_w1 = p->g_w1; a.g_b1 = p->g_b1; a.g_w2 = p->g_w2; a.g_b2 = p->g_b2;
    a.g_vol = p->g_vol; a.g_scale = p->g_scale; a.c_ss = p->c_ss; a.loss = p->loss; a.d_p = p->d_p;
    // The kernel chooses the pairs per workgroup on the device so that the chunks of all bones fit 2 workgroups per CU (its
    // resident slots) -- sum_j ceil(cnt_j / ppw) <= target + J whatever the data -- so that is all the grid ever needs: a grid
    // sized from the row CAPACITY launched thousands of workgroups that found no work (~90 us of empty launches per step).
    const int target = 2 * num_cu() - J;
    const long by_cap = ((long)p->rows_cap * J + AB_PPW_MIN - 1) / AB_PPW_MIN + J;      // no more chunks than that can exist
    const long grid = by_cap < target + J ? by_cap : target + J;
    hipLaunchKernelGGL(k_assign_bwd, dim3((unsigned)grid), dim3(AB_THREADS), 0, (hipStream_t)stream, a, target);
    DANBO_LAUNCH_RET();
}
