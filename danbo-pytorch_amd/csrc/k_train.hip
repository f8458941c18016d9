// danbo_train_step: forward, losses and backward of one DANBO training batch behind ONE C call.
// Reference: Trainer.train_batch (core/trainer.py:257-302) = render(...) [RayCaster.render_rays in training mode,
// core/raycasters.py:245-377] -> compute_loss (:348-394, 396-422, 507-553) -> loss.backward() (:563-576).
// Host code only: it enqueues the kernels of the other translation units on `stream` (about 75 launches, no allocation, no
// synchronisation, every size that depends on the data is read from device counters) -- so the whole step can be captured
// in a HIP graph.  The optimizer update is danbo_adam_step on the flat buffers (after the gradient all-reduce, if any).
//
// Structure of a step (R rays of G poses, S coarse + Sf importance samples, one network):
//   pack         fp16 hi/lo fragments of every MLP matrix in both orientations (they changed in the last Adam step)
//   geometry     cylinder / per-bone-box bounds, stratified depths, pose GNN -> volumes (activations kept), per-ray view inputs
//   pass 0       K1a cull+compact -> K1b+K2 (h rows) -> PE rows -> 10 dense layers on the compacted rows (+ R empty-space rows)
//                -> colour head -> composite -> importance depths
//   pass 1       the same network on the in-volume importance samples -> composite of the merged samples
//   loss         L1 / MSE gradients of both passes -> composite adjoints -> un-merge -> rows
//   backward     ONE sweep over the rows of both passes: 10 input-gradient GEMMs (transposed packings, ReLU bits recorded by
//                the forward), frame-code gradients, PE adjoint, K2/K1b adjoint by (row, valid bone) pairs with the forward
//                recomputed, pose GNN adjoint, volume-scale term -- and LAST all weight gradients in one grouped launch
//                (data-parallel training reduces everything else under it, danbo_train_step_phase)
// The trunk's activations and their gradients live in k_linear16's fragment order between the layers (carve, fwd_frag / bwd_frag).
#include <limits.h>
#include <stdlib.h>
#include "common.hpp"

using namespace danbo;

namespace {

struct Carver {
    char* base;
    size_t used;
    template <class T>
    T* take(size_t n) {
        used = (used + 255) & ~(size_t)255;
        T* p = base ? reinterpret_cast<T*>(base + used) : nullptr;
        used += n * sizeof(T);
        return p;
    }
};

constexpr int N_FWD = 10, N_BWD = 10, N_MAT = N_FWD + N_BWD;   // packed matrices: trunk 0..7, fa, view | view^T, fa^T, trunk 7..0 ^T
constexpr int LD_PE = 196, LD_VIN = 156, LD_FA = 260, LD_VF = 412, LD_X5 = 452;
// running max |.| slots (power-of-two pre-scales of the gradient GEMMs)
enum { MX_RAW = 0, MX_V, MX_VF, MX_Z7, MX_Z6, MX_Z5, MX_Z4, MX_Z3, MX_Z2, MX_Z1, MX_Z0, MX_X0, MX_COUNT };

struct TrainBuffers {
    // zeroed at the start of every step (one memset)
    char* zero_begin;
    int32_t *cnt, *cntb;
    float *maxabs, *loss, *g_vol;
    char* zero_end;
    // geometry
    float *near, *far, *cyl_scratch, *z_c, *z_f, *z_sorted, *vol_scratch, *volumes, *vin, *adj_prod;
    int32_t* order;
    uint32_t *bits_c, *bits_f;
    // rows
    int32_t *row_sample, *row_ray, *lists;
    float *h_rows, *pe, *vinr, *y[8], *fa, *hv, *raw_rows, *raw_c, *raw_f, *raw_empty, *raw_sorted;
    uint2* relu[8];
    // composite outputs that are not handed to the caller
    float *weights0;
    // backward
    float *g_rgb, *g_acc, *g_rgb0, *g_acc0, *d_raw_c, *d_raw_f, *d_raw_sorted, *d_raw_rows, *dpre_v, *d_alpha4, *d_vfeat, *dz[8], *d_x5,
        *d_x0, *d_h, *pose_bwd_scratch, *dw_scratch;
    uint8_t *label_c, *label_f;
    // packing
    char* packed;
    float *wmax, *wscale_inv;
    void* assign16;
};

struct Shapes {
    int R, G, S, Sf, chunk, Wg, n_codes;
    long rows_cap;
};

void describe_mats(const DanboTrainModel* m, DanboPackDesc* d) {
    const int NONE = INT_MAX;
    auto lin = [&](const float* w, int N, int K) { return DanboPackDesc{w, nullptr, K, 1, 0, 0, N, K, 0, 0, NONE, NONE}; };
    auto lin_t = [&](const float* w, int N_out /*= K of the layer*/, int K_in /*= N of the layer*/) {
        return DanboPackDesc{w, nullptr, 1, N_out, 0, 0, N_out, K_in, 0, 0, NONE, NONE};
    };
    const float* const* pw = &m->p[DANBO_T_PTS_W0];
    // ---- forward
    d[0] = lin(pw[0], 256, 195);
    for (int l = 1; l < 8; ++l) d[l] = lin(pw[l], 256, 256);
    d[5] = DanboPackDesc{pw[5], nullptr, 451, 1, 0, 0, 256, 195, 256, 0, NONE, NONE};          // [pe | y4]
    d[8] = DanboPackDesc{m->p[DANBO_T_FEAT_W], m->p[DANBO_T_ALPHA_W], 256, 1, 256, 1, 257, 256, 0, 0, 256, NONE};   // feature rows, then alpha
    d[9] = DanboPackDesc{m->p[DANBO_T_VIEWS_W], nullptr, 256 + m->view_ch, 1, 0, 0, 128, 256, m->view_ch, 0, NONE, NONE};   // [feature | view inputs]
    // ---- backward: y = dz W, i.e. "weight" W^T [K_layer, N_layer]
    d[10] = lin_t(m->p[DANBO_T_VIEWS_W], 256 + m->view_ch, 128);
    d[10].sk = 256 + m->view_ch;   // W^T[n', k'] = views_w[k' * (256 + Cv) + n']
    d[10].sn = 1;
    // fa^T: inputs [d feature (256) | d alpha (1)] -> 256 outputs: W'[n', k'] = k' < 256 ? feature_w[k' * 256 + n'] : alpha_w[n']
    d[11] = DanboPackDesc{m->p[DANBO_T_FEAT_W], m->p[DANBO_T_ALPHA_W], 1, 256, 1, 256, 256, 256, 1, 0, NONE, 256};
    for (int l = 7; l >= 1; --l) {
        DanboPackDesc& t = d[12 + (7 - l)];
        if (l == 5) t = DanboPackDesc{pw[5], nullptr, 1, 451, 0, 0, 451, 256, 0, 195, NONE, NONE};   // outputs [d y4 (256) | d pe (195)]
        else t = DanboPackDesc{pw[l], nullptr, 1, 256, 0, 0, 256, 256, 0, 0, NONE, NONE};
    }
    d[19] = DanboPackDesc{pw[0], nullptr, 1, 195, 0, 0, 195, 256, 0, 0, NONE, NONE};
    // which inputs arrive in fragment order (see carve): forward y_{l-1} (the skip layer: its second input y4); backward dz_l
    // for l = 7, 6, 3, 2, 1 and dz_0
    for (int l = 1; l < 8; ++l) d[l].frag_in = l == 5 ? 2 : 1;
    for (int l = 7; l >= 1; --l) d[12 + (7 - l)].frag_in = (l == 5 || l == 4) ? 0 : 1;
    d[19].frag_in = 1;
}

// DanboLinearEx.frag of the forward layer l and of the adjoint step that consumes dz_l (l = 8: the feature/alpha adjoint -> dz7)
inline int fwd_frag(int l) { return l == 0 ? 4 : l == 5 ? 6 : l == 7 ? 1 : 5; }
inline int bwd_frag(int l) { return l == 8 ? 4 : l == 7 ? 5 : l == 6 ? 1 : l == 5 ? 0 : l == 4 ? 4 : l >= 1 ? 5 : 1; }
inline bool dz_is_frag(int l) { return l != 5 && l != 4; }

TrainBuffers carve(Carver& c, const Shapes& s, const DanboTrainModel* m, long packed_bytes, long dw_floats) {
    TrainBuffers b;
    const size_t Mc = (size_t)s.R * s.S, Mf = (size_t)s.R * s.Sf, n = (size_t)s.rows_cap;
    // ---- zero block
    c.used = (c.used + 255) & ~(size_t)255;
    b.zero_begin = c.base ? c.base + c.used : nullptr;
    b.cnt = c.take<int32_t>(8);
    b.cntb = c.take<int32_t>(J);
    b.maxabs = c.take<float>(32);
    b.loss = c.take<float>(8);
    b.g_vol = c.take<float>((size_t)s.G * J * VOL);
    b.zero_end = c.base ? c.base + c.used : (char*)c.used;
    // ---- geometry
    b.near = c.take<float>(s.R);
    b.far = c.take<float>(s.R);
    b.cyl_scratch = c.take<float>(8 * (size_t)((s.R + s.chunk - 1) / s.chunk));
    b.z_c = c.take<float>(Mc);
    b.z_f = c.take<float>(Mf);
    b.z_sorted = c.take<float>(Mc + Mf);
    b.order = c.take<int32_t>(Mc + Mf);
    b.vol_scratch = c.take<float>(3 * (size_t)s.G * J * s.Wg);
    b.volumes = c.take<float>((size_t)s.G * J * VOL);
    b.vin = c.take<float>((size_t)s.R * LD_VIN);
    b.adj_prod = c.take<float>(3 * J * J);
    b.bits_c = c.take<uint32_t>(Mc);
    b.bits_f = c.take<uint32_t>(Mf);
    // ---- rows
    b.row_sample = c.take<int32_t>(n);
    b.row_ray = c.take<int32_t>(n);
    b.lists = c.take<int32_t>((size_t)J * n);
    b.h_rows = c.take<float>(n * 16);
    b.pe = c.take<float>(n * LD_PE);
    b.vinr = c.take<float>(n * LD_VIN);
    // trunk activations y0 .. y6 and their gradients dz7, dz6, dz3 .. dz0 live in k_linear16's fragment order between the
    // layers (rows padded to whole 128-row tiles, + one tile of slack for the weight-gradient kernel's last step); y7, dz5 and
    // dz4 = [d y4 | d pe] stay row-major for their other consumers (the 257- and 451-wide layers, the PE adjoint)
    const size_t nf = (n + 127) / 128 * 128 + 128;
    for (int l = 0; l < 8; ++l) { b.y[l] = c.take<float>((l < 7 ? nf : n) * 256); b.relu[l] = c.take<uint2>(n * 4); }
    b.fa = c.take<float>(n * LD_FA);
    b.hv = c.take<float>(n * 128);
    b.raw_rows = c.take<float>(n * 4);
    b.raw_c = c.take<float>(Mc * 4);
    b.raw_f = c.take<float>(Mf * 4);
    b.raw_empty = c.take<float>((size_t)s.R * 4);
    b.raw_sorted = c.take<float>((Mc + Mf) * 4);
    b.weights0 = c.take<float>(Mc);
    // ---- backward
    b.g_rgb = c.take<float>((size_t)s.R * 3);
    b.g_acc = c.take<float>(s.R);
    b.g_rgb0 = c.take<float>((size_t)s.R * 3);
    b.g_acc0 = c.take<float>(s.R);
    b.d_raw_c = c.take<float>(Mc * 4);
    b.d_raw_f = c.take<float>(Mf * 4);
    b.d_raw_sorted = c.take<float>((Mc + Mf) * 4);
    b.d_raw_rows = c.take<float>(n * 4);
    b.dpre_v = c.take<float>(n * 128);
    b.d_alpha4 = c.take<float>(n * 4);
    b.d_vfeat = c.take<float>(n * LD_VF);
    for (int l = 0; l < 8; ++l) b.dz[l] = l == 4 ? nullptr : c.take<float>((l == 5 ? n : nf) * 256);
    b.d_x5 = c.take<float>(n * LD_X5);
    b.dz[4] = b.d_x5;                                   // [d y4 (masked: dz4) | d pe]
    b.d_x0 = c.take<float>(n * LD_PE);
    b.d_h = c.take<float>(n * 16);
    b.pose_bwd_scratch = c.take<float>(2 * (size_t)s.G * J * s.Wg);
    b.dw_scratch = c.take<float>(dw_floats);
    b.label_c = c.take<uint8_t>(Mc);
    b.label_f = c.take<uint8_t>(Mf);
    // ---- packing
    b.packed = c.take<char>(packed_bytes);
    b.wmax = c.take<float>(N_MAT);
    b.wscale_inv = c.take<float>(N_MAT);
    b.assign16 = c.take<char>(DANBO_ASSIGN16_PACKED_BYTES);
    (void)m;
    return b;
}

constexpr int N_DW = 13;           // 12 parameter matrices, the skip layer's as two (see describe_dw)
constexpr int DW_SLICES = 10;      // 24 tiles of 128 x 256 x 10 row slices = 240 workgroups: one per CU

void describe_dw(const DanboTrainModel* m, const TrainBuffers& b, DanboDwLayer* L) {
    const int mx_of_dz[8] = {MX_Z0, MX_Z1, MX_Z2, MX_Z3, MX_Z4, MX_Z5, MX_Z6, MX_Z7};
    for (int l = 0; l < 8; ++l) {
        DanboDwLayer& d = L[l];
        d = DanboDwLayer{};
        d.dy = b.dz[l];
        d.ldy = l == 4 ? LD_X5 : 256;
        d.N = 256;
        d.dy_maxabs = b.maxabs + mx_of_dz[l];
        d.gw = m->g[DANBO_T_PTS_W0 + l];
        d.gb = m->g[DANBO_T_PTS_B0 + l];
        d.frag = dz_is_frag(l) ? 1 : 0;
        if (l == 0) { d.x1 = b.pe; d.ld1 = LD_PE; d.K1 = 195; }
        else if (l == 5) { d.x1 = b.pe; d.ld1 = LD_PE; d.K1 = 195; d.gw_ld = 451; d.gw_col0 = 0; }   // [pe | y4]: the pe columns ...
        else { d.x1 = b.y[l - 1]; d.ld1 = 256; d.K1 = 256; d.frag |= 2; }
    }
    {   // ... and the y4 columns of pts_linears.5.weight as a layer of their own: one input layout per layer
        DanboDwLayer& d = L[12];
        d = L[5];
        d.x1 = b.y[4]; d.ld1 = 256; d.K1 = 256; d.frag |= 2; d.gw_col0 = 195; d.gb = nullptr;
    }
    DanboDwLayer& f = L[8];      // feature_linear
    f = DanboDwLayer{};
    f.dy = b.d_vfeat; f.ldy = LD_VF; f.N = 256; f.dy_maxabs = b.maxabs + MX_VF;
    f.x1 = b.y[7]; f.ld1 = 256; f.K1 = 256; f.gw = m->g[DANBO_T_FEAT_W]; f.gb = m->g[DANBO_T_FEAT_B];
    DanboDwLayer& al = L[9];     // alpha_linear
    al = DanboDwLayer{};
    al.dy = b.d_alpha4; al.ldy = 4; al.N = 1; al.dy_maxabs = b.maxabs + MX_VF;
    al.x1 = b.y[7]; al.ld1 = 256; al.K1 = 256; al.gw = m->g[DANBO_T_ALPHA_W]; al.gb = m->g[DANBO_T_ALPHA_B];
    DanboDwLayer& v = L[10];     // views_linears.0
    v = DanboDwLayer{};
    v.dy = b.dpre_v; v.ldy = 128; v.N = 128; v.dy_maxabs = b.maxabs + MX_V;
    v.x1 = b.fa; v.ld1 = LD_FA; v.K1 = 256; v.x2 = b.vinr; v.ld2 = LD_VIN; v.K2 = m->view_ch;
    v.gw = m->g[DANBO_T_VIEWS_W]; v.gb = m->g[DANBO_T_VIEWS_B];
    DanboDwLayer& r = L[11];     // rgb_linear
    r = DanboDwLayer{};
    r.dy = b.d_raw_rows; r.ldy = 4; r.N = 3; r.dy_maxabs = b.maxabs + MX_RAW;
    r.x1 = b.hv; r.ld1 = 128; r.K1 = 128; r.gw = m->g[DANBO_T_RGB_W]; r.gb = m->g[DANBO_T_RGB_B];
}

bool model_ok(const DanboTrainModel* m) {
    if (!m) return false;
    for (int i = 0; i < DANBO_T_COUNT; ++i) {
        const bool optional = i == DANBO_T_CODES;
        if ((!m->p[i] || !m->g[i]) && !(optional && m->n_codes == 0)) return false;
    }
    if (!m->g_adj0 || !m->g_adj1 || !m->a_adj || !m->align || !m->init_scale || !m->g_flat) return false;
    if (m->p[DANBO_T_ALPHA_B] != m->p[DANBO_T_FEAT_B] + 256) return false;   // evaluated as one 257-wide layer: contiguous biases
    const int nd = 3 * (1 + 2 * m->L_view);
    if (m->view_ch != nd + (m->n_codes > 0 ? m->code_size : 0) || m->view_ch > LD_VIN - 1) return false;
    if (m->L_voxel != 6 || m->graph_width < 1 || m->graph_width > 256) return false;      // 195 = 15 (1 + 2 * 6) input columns
    return true;
}

// adjacency products adj_w * adj for the two graph layers and the assignment net, and the volume-scale loss + gradient
// (reference trainer.py:538-553: penalty * sum_j prod_k max(|s_jk|, 0.05 init_jk))
__global__ __launch_bounds__(256) void k_train_small(const float* aw0, const float* a0, const float* aw1, const float* a1, const float* aw2,
                                                     const float* a2, float* prod, const float* scale, const float* init_scale,
                                                     float penalty, float* g_scale, float* loss) {
    for (int i = threadIdx.x; i < J * J; i += 256) {
        prod[i] = aw0[i] * a0[i];
        prod[J * J + i] = aw1[i] * a1[i];
        prod[2 * J * J + i] = aw2[i] * a2[i];
    }
    float term = 0.f;
    if (threadIdx.x < J && penalty != 0.f) {
        const int j = threadIdx.x;
        float c[3];
        bool live[3];
        for (int k = 0; k < 3; ++k) {
            const float s = fabsf(scale[3 * j + k]), lo = init_scale[3 * j + k] * 0.05f;
            live[k] = s >= lo;            // clamp(min = lo): the gradient passes where the value is not below the bound
            c[k] = live[k] ? s : lo;
        }
        term = c[0] * c[1] * c[2] * penalty;
        for (int k = 0; k < 3; ++k) {
            const float others = c[(k + 1) % 3] * c[(k + 2) % 3];
            const float sg = scale[3 * j + k] > 0.f ? 1.f : (scale[3 * j + k] < 0.f ? -1.f : 0.f);
            if (live[k]) atomicAdd(g_scale + 3 * j + k, penalty * others * sg);
        }
    }
    if (threadIdx.x < 64) {
        term = wave_total(term);
        if (threadIdx.x == 0) loss[3] = term;
    }
}

// raw[m] = raw_empty[ray] for the samples K3's scatter never wrote (S > 64: the unfused composite reads a dense tensor)
__global__ __launch_bounds__(256) void k_fill_raw_lazy(const float4* __restrict__ raw_empty, const uint32_t* __restrict__ bits, int R, int S,
                                                       float4* __restrict__ raw) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (long)R * S; i += (long)gridDim.x * blockDim.x)
        if (bits[i] == 0u) raw[i] = raw_empty[i / S];
}

#define DANBO_TRY(call) do { const int rc_ = (call); if (rc_ != 0) return rc_; } while (0)

}  // namespace

extern "C" size_t danbo_train_workspace(const DanboTrainModel* m, int R, int G, int S, int Sf, int chunk) {
    if (!model_ok(m) || R < 1 || G < 1 || S < 3 || Sf < 1 || chunk < 1) return 0;
    DanboPackDesc d[N_MAT];
    describe_mats(m, d);
    const long packed_bytes = danbo_linear16_group_bytes(d, N_MAT);
    Shapes s{R, G, S, Sf, chunk, m->graph_width, m->n_codes, (long)R * (S + Sf + 1)};
    Carver c{nullptr, 0};
    TrainBuffers b0 = carve(c, s, m, packed_bytes, 0);
    DanboDwLayer L[N_DW];
    describe_dw(m, b0, L);
    const long dw = danbo_dw16_scratch_floats(L, N_DW, DW_SLICES);
    Carver c2{nullptr, 0};
    carve(c2, s, m, packed_bytes, dw);
    return c2.used + 512;
}

// phase 0: the whole step; 1: everything up to and including the pose-GNN adjoint -- from then on every gradient except the
// dense layers' (pts_linears.*, alpha / feature / views / rgb_linear) is final; 2: the dense layers' weight gradients and the
// loss copy.  Data-parallel training launches the all-reduce of the finished part (pose GNN, assignment net, axis scales: the
// first tensors of the flat buffer, 7 of its 10 MB) on a side stream between phases 1 and 2.
static int train_step_impl(const DanboTrainModel* m, const DanboTrainBatch* bt, const DanboTrainOut* o, void* workspace,
                           size_t workspace_bytes, void* stream, int phase) {
    DANBO_CHECK_ARG(model_ok(m) && bt && o && workspace && phase >= 0 && phase <= 2);
    const int R = bt->R, G = bt->G, S = bt->S, Sf = bt->Sf;
    DANBO_CHECK_ARG(R >= 1 && G >= 1 && R % G == 0 && S >= 3 && Sf >= 1 && S <= 256 && S + Sf <= 256 && bt->chunk >= 1);
    DANBO_CHECK_ARG(bt->rays_o && bt->rays_d && bt->skts && bt->bones && bt->cyls && bt->target);
    DANBO_CHECK_ARG(m->n_codes == 0 || bt->cam_idx);
    DANBO_CHECK_ARG(o->rgb_map && o->disp_map && o->acc_map && o->alpha && o->weights && o->rgb0 && o->disp0 && o->acc0 && o->alpha0 && o->loss);
    DANBO_CHECK_ARG(workspace_bytes >= danbo_train_workspace(m, R, G, S, Sf, bt->chunk));
    hipStream_t st = (hipStream_t)stream;

    DanboPackDesc desc[N_MAT];
    describe_mats(m, desc);
    const long packed_bytes = danbo_linear16_group_bytes(desc, N_MAT);
    Shapes sh{R, G, S, Sf, bt->chunk, m->graph_width, m->n_codes, (long)R * (S + Sf + 1)};
    DanboDwLayer dwl[N_DW];
    {
        Carver c0{nullptr, 0};
        TrainBuffers b0 = carve(c0, sh, m, packed_bytes, 0);
        describe_dw(m, b0, dwl);
    }
    const long dw_floats = danbo_dw16_scratch_floats(dwl, N_DW, DW_SLICES);
    Carver c{reinterpret_cast<char*>(((uintptr_t)workspace + 255) & ~(uintptr_t)255), 0};
    const TrainBuffers b = carve(c, sh, m, packed_bytes, dw_floats);
    describe_dw(m, b, dwl);
    const int ncap = (int)sh.rows_cap;
    const float B = m->density_scale;

    // dev aid: DANBO_TRAIN_STOP_AFTER=<stage> makes the call return after that stage (bisecting a fault inside a captured graph)
    // (the environment is read ONCE per process, not per step)
    static const int stop_after = [] { const char* e = getenv("DANBO_TRAIN_STOP_AFTER"); return e ? atoi(e) : 1000; }();
#define DANBO_STAGE(n) do { if (stop_after <= (n)) { DANBO_LAUNCH_RET(); } } while (0)
    if (phase != 2) {
    // ---- zero: counters, running maxima, loss terms, volume gradients; the flat parameter gradient
    zero_words(b.zero_begin, (long)((b.zero_end - b.zero_begin) / 4), m->g_flat, (long)m->n_flat, st);

    // ---- packings of the current weights; adjacency products; volume-scale loss
    long off[N_MAT];
    DANBO_TRY(danbo_linear16_pack_group(desc, N_MAT, b.packed, off, b.wmax, b.wscale_inv, stream));
    hipLaunchKernelGGL(k_train_small, dim3(1), dim3(256), 0, st, m->p[DANBO_T_G_ADJW0], m->g_adj0, m->p[DANBO_T_G_ADJW1], m->g_adj1,
                       m->p[DANBO_T_A_ADJW], m->a_adj, b.adj_prod, m->p[DANBO_T_AXIS_SCALE], m->init_scale, m->vol_scale_penalty,
                       m->g[DANBO_T_AXIS_SCALE], b.loss);
    const float* adjw0 = b.adj_prod;
    const float* adjw1 = b.adj_prod + J * J;
    const float* adjw_a = b.adj_prod + 2 * J * J;
    DANBO_TRY(danbo_assign16_pack(m->p[DANBO_T_A_W0], adjw_a, m->p[DANBO_T_A_W1], b.assign16, stream));
    DANBO_STAGE(1);

    // ---- bounds, depths (reference raycasters.py:310-311), pose volumes, per-ray view inputs
    const float* axis_scale = m->p[DANBO_T_AXIS_SCALE];
    DANBO_TRY(danbo_near_far_cylinder(bt->rays_o, bt->rays_d, bt->cyls, R, G, 0.f, 1.f, bt->near_in, bt->far_in, bt->chunk, b.cyl_scratch,
                                      b.near, b.far, stream));
    if (m->use_volume_near_far)
        DANBO_TRY(danbo_near_far_boxes(bt->rays_o, bt->rays_d, bt->skts, m->align, axis_scale, R, G, b.near, b.far, stream));
    DANBO_TRY(danbo_coarse_samples(b.near, b.far, R, S, bt->t_rand, b.z_c, stream));
    DANBO_TRY(danbo_pose_volumes_fwd(bt->bones, G, m->L_graph, m->graph_width, m->p[DANBO_T_G_W0], adjw0, m->p[DANBO_T_G_B0],
                                     m->p[DANBO_T_G_W1], adjw1, m->p[DANBO_T_G_B1], m->p[DANBO_T_G_W2], m->p[DANBO_T_G_B2],
                                     m->p[DANBO_T_G_W3], m->p[DANBO_T_G_B3], b.vol_scratch, b.volumes, stream));
    DANBO_TRY(danbo_train_view_inputs(bt->rays_d, bt->skts, R, G, m->ray_mode, m->normalise, m->L_view, m->p[DANBO_T_CODES], m->n_codes,
                                      m->code_size, bt->cam_idx, b.vin, LD_VIN, stream));
    DANBO_STAGE(2);

    // ---- one network pass over the compacted rows
    bool stopped = false;
#define NET_STAGE(n) do { if (stop_after <= (n)) { stopped = true; return 0; } } while (0)
    auto network = [&](int pass) -> int {
        const float* zz = pass == 0 ? b.z_c : b.z_f;
        const int s = pass == 0 ? S : Sf;
        uint32_t* bits = pass == 0 ? b.bits_c : b.bits_f;
        const int32_t* first = pass == 0 ? nullptr : b.cnt + 2;       // first row of the pass
        const int32_t* count = pass == 0 ? b.cnt + 2 : b.cnt + 3;     // its number of rows
        DANBO_TRY(danbo_bone_cull(bt->rays_o, bt->rays_d, zz, nullptr, R, s, G, bt->skts, m->align, axis_scale, bits, b.row_sample + R, b.cnt,
                                  stream));
        NET_STAGE(21);
        DANBO_TRY(danbo_gather_assign_blend16_train(bt->rays_o, bt->rays_d, zz, R, s, G, bt->skts, m->align, axis_scale, b.volumes, bits,
                                                    b.row_sample + R, b.cnt, pass == 0 ? nullptr : b.cnt + 1, ncap - R, b.assign16,
                                                    m->p[DANBO_T_A_B0], m->p[DANBO_T_A_B1], m->p[DANBO_T_A_W2], m->p[DANBO_T_A_B2],
                                                    b.h_rows + (size_t)R * 16, stream));
        NET_STAGE(22);
        DANBO_TRY(danbo_train_rows_fwd(b.h_rows, b.row_sample, b.cnt, pass, R, s, ncap, m->L_voxel, b.vin, LD_VIN, b.pe, LD_PE, b.vinr,
                                       b.row_ray, stream));
        NET_STAGE(23);
        DanboLinearEx ex{};
        // the trunk works on whole 128-row tiles of the fragment-order buffers: the importance pass starts at the tile boundary
        // below its first row (cnt[6], cnt[7]) and recomputes the coarse rows in between, bit for bit
        ex.first = pass == 0 ? nullptr : b.cnt + 6;
        const int32_t* count_t = pass == 0 ? b.cnt + 2 : b.cnt + 7;
        for (int l = 0; l < 8; ++l) {
            ex.relu_out = b.relu[l];
            ex.wscale_inv = b.wscale_inv + l;
            ex.frag = fwd_frag(l);
            const float* x1 = l == 0 || l == 5 ? b.pe : b.y[l - 1];
            const int ld1 = l == 0 || l == 5 ? LD_PE : 256, K1 = l == 0 || l == 5 ? 195 : 256;
            DANBO_TRY(danbo_linear16_ex(x1, ld1, K1, l == 5 ? b.y[4] : nullptr, 256, l == 5 ? 256 : 0, b.packed + off[l],
                                        m->p[DANBO_T_PTS_B0 + l], 256, 1, b.y[l], 256, ncap, count_t, &ex, stream));
            NET_STAGE(24 + l);
        }
        ex.first = first;
        ex.frag = 0;
        ex.relu_out = nullptr;
        ex.wscale_inv = b.wscale_inv + 8;
        DANBO_TRY(danbo_linear16_ex(b.y[7], 256, 256, nullptr, 0, 0, b.packed + off[8], m->p[DANBO_T_FEAT_B], 257, 0, b.fa, LD_FA, ncap, count,
                                    &ex, stream));
        ex.wscale_inv = b.wscale_inv + 9;
        DANBO_TRY(danbo_linear16_ex(b.fa, LD_FA, 256, b.vinr, LD_VIN, m->view_ch, b.packed + off[9], m->p[DANBO_T_VIEWS_B], 128, 1, b.hv, 128,
                                    ncap, count, &ex, stream));
        return danbo_train_rgb_head_fwd(b.hv, b.fa, LD_FA, m->p[DANBO_T_RGB_W], m->p[DANBO_T_RGB_B], b.row_sample, b.cnt, pass, R, ncap,
                                        b.raw_rows, pass == 0 ? b.raw_c : b.raw_f, b.raw_empty, stream);
    };
    DANBO_TRY(network(0));
    if (stopped) { DANBO_LAUNCH_RET(); }
    DANBO_STAGE(3);
    if (S <= 64 && Sf <= 64) {
        DANBO_TRY(danbo_composite_importance_fwd(b.raw_c, b.raw_empty, b.bits_c, b.z_c, bt->rays_d, R, S, Sf, B, bt->noise_c, bt->u_rand, o->rgb0,
                                                 o->disp0, o->acc0, b.weights0, o->alpha0, b.z_f, b.z_sorted, b.order, stream));
    } else {
        hipLaunchKernelGGL(k_fill_raw_lazy, dim3(stream_grid((long)R * S, 256)), dim3(256), 0, st, reinterpret_cast<const float4*>(b.raw_empty),
                           b.bits_c, R, S, reinterpret_cast<float4*>(b.raw_c));
        DANBO_TRY(danbo_composite_fwd(b.raw_c, b.z_c, bt->rays_d, R, S, B, bt->noise_c, o->rgb0, o->disp0, o->acc0, b.weights0, o->alpha0, stream));
        DANBO_TRY(danbo_importance_samples(b.z_c, b.weights0, R, S, Sf, bt->u_rand, b.z_f, b.z_sorted, b.order, stream));
    }
    DANBO_STAGE(4);
    DANBO_TRY(network(1));
    DANBO_STAGE(5);
    DANBO_TRY(danbo_composite_merged_fwd(b.raw_c, b.raw_f, b.raw_empty, b.bits_c, b.bits_f, b.order, b.z_sorted, bt->rays_d, R, S, Sf, B,
                                         bt->noise_f, o->rgb_map, o->disp_map, o->acc_map, o->weights, o->alpha, b.raw_sorted, stream));

    DANBO_STAGE(6);
    // ---- losses and the adjoints of the two composites
    DANBO_TRY(danbo_train_loss_grad(o->rgb_map, o->acc_map, o->rgb0, o->acc0, bt->target, bt->bgs, m->use_background, R, m->loss_mse,
                                    m->rgb_loss_coef, m->rgb_loss_coef * m->coarse_weight, b.g_rgb, b.g_acc, b.g_rgb0, b.g_acc0, b.loss, stream));
    DANBO_TRY(danbo_composite_bwd_lazy(b.raw_c, b.raw_empty, b.bits_c, b.z_c, bt->rays_d, R, S, B, bt->noise_c, b.g_rgb0, b.g_acc0, b.d_raw_c,
                                       stream));
    DANBO_TRY(danbo_composite_bwd_lazy(b.raw_sorted, nullptr, nullptr, b.z_sorted, bt->rays_d, R, S + Sf, B, bt->noise_f, b.g_rgb, b.g_acc,
                                       b.d_raw_sorted, stream));
    DANBO_TRY(danbo_train_draw_unmerge(b.d_raw_c, b.d_raw_sorted, b.order, b.bits_c, b.bits_f, o->weights, o->alpha, R, S, Sf, b.d_raw_f,
                                       b.d_raw_rows, b.label_c, b.label_f, b.loss, b.maxabs + MX_RAW, stream));
    DANBO_TRY(danbo_train_rgb_head_bwd(b.hv, m->p[DANBO_T_RGB_W], b.d_raw_c, b.d_raw_f, b.row_sample, b.cnt, R, ncap, b.d_raw_rows, b.dpre_v,
                                       b.d_alpha4, b.maxabs + MX_V, b.maxabs + MX_VF, stream));

    DANBO_STAGE(7);
    // ---- input-gradient GEMMs over the rows of both passes
    const int32_t* all_rows = b.cnt + 4;
    {
        DanboLinearEx ex{};
        ex.in_maxabs = b.maxabs + MX_V; ex.out_maxabs = b.maxabs + MX_VF; ex.wscale_inv = b.wscale_inv + 10;
        DANBO_TRY(danbo_linear16_ex(b.dpre_v, 128, 128, nullptr, 0, 0, b.packed + off[10], nullptr, 256 + m->view_ch, 0, b.d_vfeat, LD_VF, ncap,
                                    all_rows, &ex, stream));
        ex.in_maxabs = b.maxabs + MX_VF; ex.out_maxabs = b.maxabs + MX_Z7; ex.wscale_inv = b.wscale_inv + 11;
        ex.relu_in = b.relu[7]; ex.mask_cols = 256;
        ex.frag = bwd_frag(8);
        DANBO_TRY(danbo_linear16_ex(b.d_vfeat, LD_VF, 256, b.d_alpha4, 4, 1, b.packed + off[11], nullptr, 256, 0, b.dz[7], 256, ncap, all_rows,
                                    &ex, stream));
        const int mx_of_dz[8] = {MX_Z0, MX_Z1, MX_Z2, MX_Z3, MX_Z4, MX_Z5, MX_Z6, MX_Z7};
        for (int l = 7; l >= 1; --l) {      // dz_{l-1} = (dz_l W_l) * [y_{l-1} > 0]
            ex.in_maxabs = b.maxabs + mx_of_dz[l]; ex.out_maxabs = b.maxabs + mx_of_dz[l - 1]; ex.wscale_inv = b.wscale_inv + 12 + (7 - l);
            ex.relu_in = b.relu[l - 1]; ex.mask_cols = 256;
            ex.frag = bwd_frag(l);
            const float* x = b.dz[l];
            const int ldx = l == 4 ? LD_X5 : 256;
            if (l == 5)
                DANBO_TRY(danbo_linear16_ex(x, ldx, 256, nullptr, 0, 0, b.packed + off[12 + 2], nullptr, 451, 0, b.d_x5, LD_X5, ncap, all_rows, &ex,
                                            stream));
            else
                DANBO_TRY(danbo_linear16_ex(x, ldx, 256, nullptr, 0, 0, b.packed + off[12 + (7 - l)], nullptr, 256, 0, b.dz[l - 1],
                                            l - 1 == 4 ? LD_X5 : 256, ncap, all_rows, &ex, stream));
        }
        ex.in_maxabs = b.maxabs + MX_Z0; ex.out_maxabs = b.maxabs + MX_X0; ex.wscale_inv = b.wscale_inv + 19;
        ex.relu_in = nullptr; ex.mask_cols = 0;
        ex.frag = bwd_frag(0);
        DANBO_TRY(danbo_linear16_ex(b.dz[0], 256, 256, nullptr, 0, 0, b.packed + off[19], nullptr, 195, 0, b.d_x0, LD_PE, ncap, all_rows, &ex, stream));
    }
    DANBO_STAGE(9);
    // ---- frame codes
    if (m->n_codes > 0)
        DANBO_TRY(danbo_train_code_grad(b.d_vfeat, LD_VF, 256 + 3 * (1 + 2 * m->L_view), m->code_size, b.row_ray, bt->cam_idx, b.cnt, ncap,
                                        m->n_codes, m->g[DANBO_T_CODES], stream));
    // ---- PE adjoint, K2 / K1b adjoint, pose GNN adjoint
    DANBO_TRY(danbo_train_pe_bwd(b.d_x0, LD_PE, b.d_x5, LD_X5, 256, b.h_rows, b.cnt, R, ncap, m->L_voxel, b.d_h, stream));
    DANBO_TRY(danbo_train_bone_lists(b.bits_c, b.bits_f, b.row_sample, b.cnt, R, ncap, b.lists, b.cntb, stream));
    DANBO_STAGE(10);
    DanboAssignBwd ab{};
    ab.rays_o = bt->rays_o; ab.rays_d = bt->rays_d; ab.z_c = b.z_c; ab.z_f = b.z_f; ab.skts = bt->skts; ab.align = m->align;
    ab.axis_scale = axis_scale; ab.volumes = b.volumes; ab.R = R; ab.S = S; ab.Sf = Sf; ab.G = G; ab.rows_cap = ncap;
    ab.row_sample = b.row_sample; ab.row_ray = b.row_ray; ab.cnt = b.cnt; ab.lists = b.lists; ab.cntb = b.cntb; ab.h_rows = b.h_rows;
    ab.d_h = b.d_h; ab.label_c = b.label_c; ab.label_f = b.label_f; ab.bits_c = b.bits_c; ab.bits_f = b.bits_f;
    ab.w0 = m->p[DANBO_T_A_W0]; ab.adj_w = m->p[DANBO_T_A_ADJW]; ab.adj = m->a_adj; ab.b0 = m->p[DANBO_T_A_B0]; ab.w1 = m->p[DANBO_T_A_W1];
    ab.b1 = m->p[DANBO_T_A_B1]; ab.w2 = m->p[DANBO_T_A_W2]; ab.b2 = m->p[DANBO_T_A_B2];
    ab.g_w0 = m->g[DANBO_T_A_W0]; ab.g_adj_w = m->g[DANBO_T_A_ADJW]; ab.g_b0 = m->g[DANBO_T_A_B0]; ab.g_w1 = m->g[DANBO_T_A_W1];
    ab.g_b1 = m->g[DANBO_T_A_B1]; ab.g_w2 = m->g[DANBO_T_A_W2]; ab.g_b2 = m->g[DANBO_T_A_B2];
    ab.g_vol = b.g_vol; ab.g_scale = m->g[DANBO_T_AXIS_SCALE];
    ab.c_ss = 2.0f * m->soft_softmax_coef / ((float)R * (float)(S + Sf));
    ab.loss = b.loss;
    DANBO_TRY(danbo_assign_blend_bwd(&ab, stream));
    DANBO_STAGE(11);
    DANBO_TRY(danbo_pose_volumes_bwd(bt->bones, G, m->L_graph, m->graph_width, m->p[DANBO_T_G_W0], m->p[DANBO_T_G_ADJW0], m->g_adj0,
                                     m->p[DANBO_T_G_B0], m->p[DANBO_T_G_W1], m->p[DANBO_T_G_ADJW1], m->g_adj1, m->p[DANBO_T_G_B1],
                                     m->p[DANBO_T_G_W2], m->p[DANBO_T_G_W3], b.vol_scratch, b.g_vol, m->g[DANBO_T_G_W0], m->g[DANBO_T_G_ADJW0],
                                     m->g[DANBO_T_G_B0], m->g[DANBO_T_G_W1], m->g[DANBO_T_G_ADJW1], m->g[DANBO_T_G_B1], m->g[DANBO_T_G_W2],
                                     m->g[DANBO_T_G_B2], m->g[DANBO_T_G_W3], m->g[DANBO_T_G_B3], b.pose_bwd_scratch, stream));
    DANBO_STAGE(12);
    }   // phase != 2
    if (phase == 1) { DANBO_LAUNCH_RET(); }
    // ---- weight / bias gradients of all dense layers (last: it needs nothing but the activations and their gradients, and
    //      data-parallel training hides the all-reduce of everything computed so far -- 7 of the 10 MB -- under it)
    const int32_t* all_rows2 = b.cnt + 4;
    DANBO_TRY(danbo_dw16(dwl, N_DW, ncap, all_rows2, DW_SLICES, b.dw_scratch, stream));
    // ---- loss terms for the caller: [0] rgb fine, [1] rgb coarse, [2] sum (label - q)^2, [3] volume scale, [4..6] row counters
    hipLaunchKernelGGL(k_copy_words_, dim3(1), dim3(64), 0, st, reinterpret_cast<const uint32_t*>(b.loss),
                       reinterpret_cast<uint32_t*>(o->loss), 4, reinterpret_cast<const uint32_t*>(b.cnt),
                       reinterpret_cast<uint32_t*>(o->counts), o->counts ? 8 : 0);
    DANBO_LAUNCH_RET();
}

extern "C" int danbo_train_step(const DanboTrainModel* m, const DanboTrainBatch* bt, const DanboTrainOut* o, void* workspace,
                                size_t workspace_bytes, void* stream) {
    return train_step_impl(m, bt, o, workspace, workspace_bytes, stream, 0);
}

extern "C" int danbo_train_step_phase(const DanboTrainModel* m, const DanboTrainBatch* bt, const DanboTrainOut* o, void* workspace,
                                      size_t workspace_bytes, int phase, void* stream) {
    return train_step_impl(m, bt, o, workspace, workspace_bytes, stream, phase);
}
