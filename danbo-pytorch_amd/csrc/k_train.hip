// danbo_train_step: forward, losses and backward of one DANBO training batch behind ONE C call.
// Reference: Trainer.train_batch (core/trainer.py:257-302) = render(...) [RayCaster.render_rays in training mode,
// core/raycasters.py:245-377] -> compute_loss (:348-394, 396-422, 507-553) -> loss.backward() (:563-576).
// Host code only: it enqueues the kernels of the other translation units on `stream` (about 75 launches, no allocation, no
// synchronisation, every size that depends on the data is read from device counters) -- so the whole step can be captured
// in a HIP graph.  The optimizer update is danbo_adam_step on the flat buffers (after the gradient all-reduce, if any).
//
// Structure of a step (R rays of G poses, S coarse + Sf importance samples, one network):
//   pack         fp16 hi/lo fragments of every MLP matrix in both orientations (they changed in the last Adam step), the merged
//                feature / view matrix W_fv
//   geometry     cylinder / per-bone-box bounds, stratified depths, pose GNN -> volumes (activations kept), per-ray view inputs and
//                view constants
//   pass 0       K1a cull+compact -> K1b+K2 (h rows) -> ONE fused trunk kernel over the compacted rows (+ R empty-space rows):
//                encoding, 8 layers, density / colour heads, raw scattered to the pass' dense tensor -> composite -> importance depths
//   pass 1       the same on the in-volume importance samples -> composite of the merged samples
//   loss         L1 / MSE gradients of both passes -> composite adjoints -> un-merge -> rows
//   backward     ONE fused input-gradient chain over the rows of both passes (d raw -> dz_7 .. dz_0 -> d h), the per-ray view
//                gradients, K2/K1b adjoint by (row, valid bone) pairs with the forward recomputed, pose GNN adjoint, volume-scale
//                term -- and LAST all weight gradients in one grouped launch + the chain rule of the merged feature / view
//                layer (data-parallel training reduces everything else under it, danbo_train_step_phase)
// Every activation the backward needs is written ONCE by the forward trunk kernel, in the order its lanes hold it (fragment
// order), and read once by the weight-gradient kernel; the input-gradient chain itself reads only sign bits.
#include <limits.h>
#include <stdlib.h>
#include <mutex>
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

constexpr int LD_VIN = 156;
// running max |.| slots: [0, 16) belong to the fused trunk's backward (dz_0 .. dz_7, d pre_v, d alpha), then the raw gradients
enum { MX_TRUNK = 0, MX_DPRE_V = 8, MX_DALPHA = 9, MX_RAW = 16, MX_COUNT };

struct TrainBuffers {
    // zeroed at the start of every step (one launch)
    char* zero_begin;
    int32_t *cnt, *cntb;
    uint32_t* ticket;      // K2's tile ticket (0 at launch, 0 after)
    float *maxabs, *loss, *g_vol, *wmax, *d_cview, *csum;
    char* zero_end;
    // geometry
    float *near, *far, *cyl_scratch, *z_c, *z_f, *z_sorted, *vol_scratch, *volumes, *vin, *cview, *adj_prod;
    int32_t* order;
    uint32_t *bits_c, *bits_f;
    // rows
    int32_t *row_sample, *row_ray, *lists;
    float *h_rows, *y, *pe, *hv, *raw_rows, *raw_c, *raw_f, *raw_empty, *raw_sorted;
    uint64_t* relu;
    uint32_t* hv_bits;
    long rows_pad;
    // composite outputs that are not handed to the caller
    float *weights0;
    // backward
    float *g_rgb, *g_acc, *g_rgb0, *g_acc0, *d_raw_c, *d_raw_f, *d_raw_sorted, *d_raw_rows, *dz, *dpre_v, *d_alpha4, *d_h, *pose_bwd_scratch,
        *dw_scratch, *g_wfv, *g_beff, *vg_part;
    uint8_t *label_c, *label_f;
    // packing
    char* packed;
    float *wfv, *b_eff, *winv;
    void* assign16;
};

struct Shapes {
    int R, G, S, Sf, chunk, Wg, n_codes;
    long rows_cap;
};

TrainBuffers carve(Carver& c, const Shapes& s, const DanboTrainModel* m, long dw_floats) {
    TrainBuffers b;
    const size_t Mc = (size_t)s.R * s.S, Mf = (size_t)s.R * s.Sf, n = (size_t)s.rows_cap;
    // rows of every fragment-order buffer: whole 128-row tiles + one tile of slack for the weight-gradient kernel's last step
    const size_t nf = (n + 127) / 128 * 128 + 128;
    b.rows_pad = (long)nf;
    // ---- zero block
    c.used = (c.used + 255) & ~(size_t)255;
    b.zero_begin = c.base ? c.base + c.used : nullptr;
    b.cnt = c.take<int32_t>(8);
    b.cntb = c.take<int32_t>(J);
    b.ticket = c.take<uint32_t>(4);
    b.maxabs = c.take<float>(32);
    b.loss = c.take<float>(8);
    b.g_vol = c.take<float>((size_t)s.G * J * VOL);
    b.wmax = c.take<float>(16);
    b.d_cview = c.take<float>((size_t)s.R * 128);
    b.csum = c.take<float>((size_t)(s.n_codes > 0 ? s.n_codes : 1) * 128);
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
    b.vin = c.take<float>((size_t)s.R * LD_VIN + 8);
    b.cview = c.take<float>((size_t)s.R * 128);
    b.adj_prod = c.take<float>(3 * J * J);
    b.bits_c = c.take<uint32_t>(Mc);
    b.bits_f = c.take<uint32_t>(Mf);
    // ---- rows
    b.row_sample = c.take<int32_t>(n);
    b.row_ray = c.take<int32_t>(n);
    b.lists = c.take<int32_t>((size_t)J * n);
    b.h_rows = c.take<float>(n * 16);
    b.y = c.take<float>(8 * nf * 256);
    b.pe = c.take<float>(nf * DANBO_TRUNK_PE_WIDTH);
    b.relu = c.take<uint64_t>(8 * nf * 4);
    b.hv = c.take<float>(nf * 128);
    b.hv_bits = c.take<uint32_t>(nf * 4);
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
    b.dz = c.take<float>(8 * nf * 256);
    b.dpre_v = c.take<float>(nf * 128);
    b.d_alpha4 = c.take<float>(n * 4);
    b.d_h = c.take<float>(n * 16);
    b.pose_bwd_scratch = c.take<float>(2 * (size_t)s.G * J * s.Wg);
    b.dw_scratch = c.take<float>(dw_floats);
    b.g_wfv = c.take<float>(128 * 256);
    b.g_beff = c.take<float>(128);
    b.vg_part = c.take<float>(DANBO_TRAIN_VG_PART_FLOATS);
    b.label_c = c.take<uint8_t>(Mc);
    b.label_f = c.take<uint8_t>(Mf);
    // ---- packing
    b.packed = c.take<char>(DANBO_TRUNK_PACKED_BYTES);
    b.wfv = c.take<float>(128 * 256);
    b.b_eff = c.take<float>(128);
    b.winv = c.take<float>(16);
    b.assign16 = c.take<char>(DANBO_ASSIGN16_PACKED_BYTES);
    (void)m;
    return b;
}

constexpr int N_DW = 12;           // 8 trunk matrices (the skip layer's as two), W_fv, alpha_linear, rgb_linear
constexpr int DW_SLICES = 10;      // 21 (padded: 22) tiles of 128 x 256 x 10 row slices = 210 workgroups: a k_dw16 workgroup fills its CU (two 256-register
                                   // wavefronts per SIMD), the CUs left over run the pose-GNN adjoint beside it (8 .. 12 slices measured: 1.71 / 1.67 / 1.65 / 1.66 ms per step)

void describe_dw(const DanboTrainModel* m, const TrainBuffers& b, DanboDwLayer* L) {
    const size_t lstride = (size_t)b.rows_pad * 256;
    auto y_of = [&](int l) { return b.y ? b.y + l * lstride : nullptr; };
    auto dz_of = [&](int l) { return b.dz ? b.dz + l * lstride : nullptr; };
    int k = 0;
    for (int l = 0; l < 8; ++l) {
        DanboDwLayer d = DanboDwLayer{};
        d.dy = dz_of(l); d.ldy = 256; d.N = 256; d.dy_maxabs = b.maxabs + MX_TRUNK + l;
        d.gw = m->g[DANBO_T_PTS_W0 + l]; d.gb = m->g[DANBO_T_PTS_B0 + l];
        d.frag = 3;                                  // both operands in fragment order
        if (l == 0 || l == 5) {                      // the encoding: 224 slots -> 195 columns
            d.x1 = b.pe; d.ld1 = DANBO_TRUNK_PE_WIDTH; d.K1 = DANBO_TRUNK_PE_WIDTH; d.x1_pe = 1;
            d.gw_ld = l == 0 ? 195 : 451; d.gw_col0 = 0;
            L[k++] = d;
            if (l == 5) {                            // ... and the y4 columns of pts_linears.5.weight as a layer of their own
                DanboDwLayer e = d;
                e.x1 = y_of(4); e.ld1 = 256; e.K1 = 256; e.x1_pe = 0; e.gw_col0 = 195; e.gb = nullptr;
                L[k++] = e;
            }
        } else {
            d.x1 = y_of(l - 1); d.ld1 = 256; d.K1 = 256;
            L[k++] = d;
        }
    }
    DanboDwLayer& f = L[k++];     // merged feature / view layer: d W_fv, d b_eff (pulled back to the two layers by danbo_train_head_chain)
    f = DanboDwLayer{};
    f.dy = b.dpre_v; f.ldy = 128; f.N = 128; f.dy_maxabs = b.maxabs + MX_DPRE_V; f.frag = 3;
    f.x1 = y_of(7); f.ld1 = 256; f.K1 = 256; f.gw = b.g_wfv; f.gb = b.g_beff;
    DanboDwLayer& al = L[k++];    // alpha_linear
    al = DanboDwLayer{};
    al.dy = b.d_alpha4; al.ldy = 4; al.N = 1; al.dy_maxabs = b.maxabs + MX_DALPHA; al.frag = 2;
    al.x1 = y_of(7); al.ld1 = 256; al.K1 = 256; al.gw = m->g[DANBO_T_ALPHA_W]; al.gb = m->g[DANBO_T_ALPHA_B];
    DanboDwLayer& r = L[k++];     // rgb_linear
    r = DanboDwLayer{};
    r.dy = b.d_raw_rows; r.ldy = 4; r.N = 3; r.dy_maxabs = b.maxabs + MX_RAW; r.frag = 2;
    r.x1 = b.hv; r.ld1 = 128; r.K1 = 128; r.gw = m->g[DANBO_T_RGB_W]; r.gb = m->g[DANBO_T_RGB_B];
}

void describe_trunk(const DanboTrainModel* m, const TrainBuffers& b, const Shapes& s, const float* d_raw_c, const float* d_raw_f,
                    DanboTrunkWeights* w, DanboTrunkRows* r) {
    *w = DanboTrunkWeights{};
    for (int l = 0; l < 8; ++l) { w->pts_w[l] = m->p[DANBO_T_PTS_W0 + l]; w->pts_b[l] = m->p[DANBO_T_PTS_B0 + l]; }
    w->alpha_w = m->p[DANBO_T_ALPHA_W]; w->alpha_b = m->p[DANBO_T_ALPHA_B]; w->feature_w = m->p[DANBO_T_FEAT_W];
    w->feature_b = m->p[DANBO_T_FEAT_B]; w->views_w = m->p[DANBO_T_VIEWS_W]; w->views_b = m->p[DANBO_T_VIEWS_B];
    w->rgb_w = m->p[DANBO_T_RGB_W]; w->rgb_b = m->p[DANBO_T_RGB_B]; w->view_ch = m->view_ch;
    w->packed = b.packed; w->wfv = b.wfv; w->b_eff = b.b_eff; w->wmax = b.wmax; w->winv = b.winv;
    *r = DanboTrunkRows{};
    r->cnt = b.cnt; r->row_sample = b.row_sample; r->h_rows = b.h_rows; r->cview = b.cview;
    r->R = s.R; r->S = s.S; r->Sf = s.Sf; r->rows_cap = (int)s.rows_cap; r->rows_pad = b.rows_pad;
    r->y = b.y; r->pe = b.pe; r->relu = b.relu; r->hv = b.hv; r->hv_bits = b.hv_bits; r->raw_rows = b.raw_rows;
    r->raw_c = b.raw_c; r->raw_f = b.raw_f; r->raw_empty = b.raw_empty; r->row_ray = b.row_ray;
    r->d_raw_c = d_raw_c; r->d_raw_f = d_raw_f; r->d_raw_rows = b.d_raw_rows; r->dz = b.dz; r->dpre_v = b.dpre_v;
    r->d_alpha4 = b.d_alpha4; r->d_h = b.d_h; r->maxabs = b.maxabs + MX_TRUNK;
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

// Independent branches of the step run on side streams (fork: the side stream waits for an event recorded on the caller's
// stream; join: the caller's stream waits for the side stream's event).  Inside a stream capture these cross-stream waits become
// the edges of the HIP graph, so a replay runs the branches concurrently.  One set per device, created on first use -- which
// has to be an eager call (the trainer warms the step up before it captures it): streams cannot be created during a capture.
struct SideStreams {
    hipStream_t s[2];
    hipEvent_t fork, join[2], mid;
    bool ok;
};
static SideStreams* side_streams() {
    static SideStreams per_dev[64];
    static bool made[64];
    static std::mutex mu;          // two host threads stepping on the same device must not both create (or half-see) the set
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    SideStreams& ss = per_dev[dev & 63];
    if (!made[dev & 63]) {
        // (a low-priority stream for the weight-gradient kernel, so that the K2 adjoint gets the compute units first, was measured:
        //  2.5 instead of 1.85 ms per step -- plain streams)
        ss.ok = hipStreamCreateWithFlags(&ss.s[0], hipStreamNonBlocking) == hipSuccess &&
                hipStreamCreateWithFlags(&ss.s[1], hipStreamNonBlocking) == hipSuccess &&
                hipEventCreateWithFlags(&ss.fork, hipEventDisableTiming) == hipSuccess &&
                hipEventCreateWithFlags(&ss.join[0], hipEventDisableTiming) == hipSuccess &&
                hipEventCreateWithFlags(&ss.join[1], hipEventDisableTiming) == hipSuccess &&
                hipEventCreateWithFlags(&ss.mid, hipEventDisableTiming) == hipSuccess;
        if (!ss.ok) (void)hipGetLastError();
        made[dev & 63] = true;
    }
    return ss.ok ? &ss : nullptr;
}

// Every exit path of the step -- the error returns included -- leaves the side streams joined: work forked onto them and never
// joined would race with the caller's next step on the shared workspace (eager mode) or leave a capture unjoined.
struct ForkGuard {
    SideStreams* ss = nullptr;
    hipStream_t st = nullptr;
    bool pending[2] = {false, false};
    int join(int i) {
        if (!ss || !pending[i]) return 0;
        pending[i] = false;
        if (hipEventRecord(ss->join[i], ss->s[i]) != hipSuccess || hipStreamWaitEvent(st, ss->join[i], 0) != hipSuccess) return (int)hipGetLastError();
        return 0;
    }
    ~ForkGuard() { (void)join(0); (void)join(1); }
};

}  // namespace

extern "C" size_t danbo_train_workspace(const DanboTrainModel* m, int R, int G, int S, int Sf, int chunk) {
    if (!model_ok(m) || R < 1 || G < 1 || S < 3 || Sf < 1 || chunk < 1) return 0;
    Shapes s{R, G, S, Sf, chunk, m->graph_width, m->n_codes, (long)R * (S + Sf + 1)};
    Carver c{nullptr, 0};
    TrainBuffers b0 = carve(c, s, m, 0);
    DanboDwLayer L[N_DW];
    describe_dw(m, b0, L);
    const long dw = danbo_dw16_scratch_floats(L, N_DW, DW_SLICES);
    Carver c2{nullptr, 0};
    carve(c2, s, m, dw);
    return c2.used + 512;
}

extern "C" int danbo_train_workspace_view(const DanboTrainModel* m, int R, int G, int S, int Sf, int chunk, void* workspace, DanboTrainView* v) {
    DANBO_CHECK_ARG(model_ok(m) && workspace && v && R >= 1 && G >= 1 && S >= 3 && Sf >= 1 && chunk >= 1);
    Shapes sh{R, G, S, Sf, chunk, m->graph_width, m->n_codes, (long)R * (S + Sf + 1)};
    Carver c{reinterpret_cast<char*>(((uintptr_t)workspace + 255) & ~(uintptr_t)255), 0};
    const TrainBuffers b = carve(c, sh, m, 0);      // (the weight-gradient scratch lies behind everything named here)
    v->z_coarse = b.z_c; v->z_fine = b.z_f; v->z_sorted = b.z_sorted; v->order = b.order; v->bits_coarse = b.bits_c; v->bits_fine = b.bits_f;
    return 0;
}

// phase 0: the whole step; 1: everything up to and including the pose-GNN adjoint -- from then on every gradient except the
// dense layers' (pts_linears.*, alpha / feature / views / rgb_linear) and the frame codes' is final; 2: the dense layers' weight
// gradients and the loss copy.  Data-parallel training launches the all-reduce of the finished part (pose GNN, assignment net,
// axis scales: the first tensors of the flat buffer, 7 of its 10 MB) on a side stream between phases 1 and 2.
static int train_step_impl(const DanboTrainModel* m, const DanboTrainBatch* bt, const DanboTrainOut* o, void* workspace,
                           size_t workspace_bytes, void* stream, int phase) {
    DANBO_CHECK_ARG(model_ok(m) && bt && o && workspace && phase >= 0 && phase <= 2);
    const int R = bt->R, G = bt->G, S = bt->S, Sf = bt->Sf;
    DANBO_CHECK_ARG(R >= 1 && G >= 1 && R % G == 0 && S >= 3 && Sf >= 1 && S <= 256 && S + Sf <= 256 && bt->chunk >= 1);
    DANBO_CHECK_ARG(bt->rays_o && bt->rays_d && bt->skts && bt->bones && bt->cyls && bt->target);
    DANBO_CHECK_ARG(m->n_codes == 0 || bt->cam_idx);
    DANBO_CHECK_ARG(bt->rng_state == nullptr || (bt->n_uniform >= 0 && bt->n_normal >= 0 && bt->n_uniform + bt->n_normal > 0 &&
                                                 (bt->n_uniform == 0 || bt->rng_uniform) && (bt->n_normal == 0 || bt->rng_normal)));
    DANBO_CHECK_ARG(o->rgb_map && o->disp_map && o->acc_map && o->alpha && o->weights && o->rgb0 && o->disp0 && o->acc0 && o->alpha0 && o->loss);
    DANBO_CHECK_ARG(workspace_bytes >= danbo_train_workspace(m, R, G, S, Sf, bt->chunk));
    hipStream_t st = (hipStream_t)stream;

    Shapes sh{R, G, S, Sf, bt->chunk, m->graph_width, m->n_codes, (long)R * (S + Sf + 1)};
    DanboDwLayer dwl[N_DW];
    {
        Carver c0{nullptr, 0};
        TrainBuffers b0 = carve(c0, sh, m, 0);
        describe_dw(m, b0, dwl);
    }
    const long dw_floats = danbo_dw16_scratch_floats(dwl, N_DW, DW_SLICES);
    Carver c{reinterpret_cast<char*>(((uintptr_t)workspace + 255) & ~(uintptr_t)255), 0};
    const TrainBuffers b = carve(c, sh, m, dw_floats);
    describe_dw(m, b, dwl);
    DanboTrunkWeights tw;
    DanboTrunkRows trw;
    describe_trunk(m, b, sh, b.d_raw_c, b.d_raw_f, &tw, &trw);
    const int ncap = (int)sh.rows_cap;
    const float B = m->density_scale;
    const int nd = 3 * (1 + 2 * m->L_view);      // direction columns of the view inputs; the frame code follows

    // dev aid: DANBO_TRAIN_STOP_AFTER=<stage> makes the call return after that stage (bisecting a fault inside a captured graph)
    // (the environment is read ONCE per process, not per step)
    static const int stop_after = dev_env("DANBO_TRAIN_STOP_AFTER", 1000);
#define DANBO_STAGE(n) do { if (stop_after <= (n)) { DANBO_LAUNCH_RET(); } } while (0)
    bool fused_tail = false, losses_reported = false;
    if (phase != 2) {
    // ---- zero: counters, running maxima, loss terms, volume / per-ray gradients; the flat parameter gradient
    zero_words(b.zero_begin, (long)((b.zero_end - b.zero_begin) / 4), m->g_flat, (long)m->n_flat, st);

    // ---- three independent prologues, concurrently:
    //   side 0: trunk weight packing -> per-ray view inputs -> view constants (needs b_eff)
    //   side 1: adjacency products + volume-scale loss -> assignment-net packing -> pose GNN (volumes)
    //   main  : ray bounds -> stratified depths -> coarse cull
    static const int fork_mask = dev_env("DANBO_TRAIN_FORK", 3);   // dev: 1 prologue, 2 backward
    SideStreams* ss = side_streams();
    void* s0 = stream;
    void* s1 = stream;
    SideStreams* const ss_all = ss;
    if (!(fork_mask & 1)) ss = nullptr;
    ForkGuard guard;
    guard.ss = ss_all;
    guard.st = st;
    if (ss) {
        if (hipEventRecord(ss->fork, st) != hipSuccess) return (int)hipGetLastError();
        if (hipStreamWaitEvent(ss->s[0], ss->fork, 0) != hipSuccess) return (int)hipGetLastError();
        guard.pending[0] = true;
        if (hipStreamWaitEvent(ss->s[1], ss->fork, 0) != hipSuccess) return (int)hipGetLastError();
        guard.pending[1] = true;
        s0 = ss->s[0];
        s1 = ss->s[1];
    }
    auto join = [&](int i) -> int { return guard.join(i); };
    const float* axis_scale = m->p[DANBO_T_AXIS_SCALE];
    // Enqueue order = the order the captured graph submits its nodes in: the caller's stream first (bounds -> depths -> cull), then
    // side 0 (trunk packing -> view inputs -> view constants: the longest chain), then side 1 (adjacency products -> assignment
    // packing -> pose GNN).  Round 4's timeline (tools/timeline_train.sh): with side 1 enqueued first, side 0's first kernel started
    // only when side 1's last one had ended (129 us into the step) and K2 at 187; in this order all three run side by side and K2
    // starts at 137.
    // ---- the step's random numbers, behind the fork (ABI 7): nothing in front of the stratified depths needs them
    if (bt->rng_state != nullptr)
        DANBO_TRY(danbo_random_draws(bt->rng_state, (long)bt->n_uniform, bt->rng_uniform, (long)bt->n_normal, bt->normal_std, bt->rng_normal, stream));
    // ---- bounds, depths (reference raycasters.py:310-311), coarse cull
    DANBO_TRY(danbo_near_far_cylinder(bt->rays_o, bt->rays_d, bt->cyls, R, G, 0.f, 1.f, bt->near_in, bt->far_in, bt->chunk, b.cyl_scratch,
                                      b.near, b.far, stream));
    if (m->use_volume_near_far)
        DANBO_TRY(danbo_near_far_boxes(bt->rays_o, bt->rays_d, bt->skts, m->align, axis_scale, R, G, b.near, b.far, stream));
    DANBO_TRY(danbo_coarse_samples(b.near, b.far, R, S, bt->t_rand, b.z_c, stream));
    DANBO_TRY(danbo_bone_cull(bt->rays_o, bt->rays_d, b.z_c, nullptr, R, S, G, bt->skts, m->align, axis_scale, nullptr, nullptr, nullptr, nullptr, b.bits_c,
                              b.row_sample + R, b.cnt,
                              stream));
    DANBO_TRY(danbo_trunk_pack(&tw, s0));
    DANBO_TRY(danbo_train_view_inputs(bt->rays_d, bt->skts, R, G, m->ray_mode, m->normalise, m->L_view, m->p[DANBO_T_CODES], m->n_codes,
                                      m->code_size, bt->cam_idx, b.vin, LD_VIN, s0));
    DANBO_TRY(danbo_train_cview(b.vin, LD_VIN, m->view_ch, m->p[DANBO_T_VIEWS_W], b.b_eff, R, b.cview, s0));
    hipLaunchKernelGGL(k_train_small, dim3(1), dim3(256), 0, (hipStream_t)s1, m->p[DANBO_T_G_ADJW0], m->g_adj0, m->p[DANBO_T_G_ADJW1], m->g_adj1,
                       m->p[DANBO_T_A_ADJW], m->a_adj, b.adj_prod, m->p[DANBO_T_AXIS_SCALE], m->init_scale, m->vol_scale_penalty,
                       m->g[DANBO_T_AXIS_SCALE], b.loss);
    const float* adjw0 = b.adj_prod;
    const float* adjw1 = b.adj_prod + J * J;
    const float* adjw_a = b.adj_prod + 2 * J * J;
    DANBO_TRY(danbo_assign16_pack(m->p[DANBO_T_A_W0], adjw_a, m->p[DANBO_T_A_W1], b.assign16, s1));
    DANBO_TRY(danbo_pose_volumes_fwd(bt->bones, G, m->L_graph, m->graph_width, m->p[DANBO_T_G_W0], adjw0, m->p[DANBO_T_G_B0],
                                     m->p[DANBO_T_G_W1], adjw1, m->p[DANBO_T_G_B1], m->p[DANBO_T_G_W2], m->p[DANBO_T_G_B2],
                                     m->p[DANBO_T_G_W3], m->p[DANBO_T_G_B3], b.vol_scratch, b.volumes, s1));
    DANBO_STAGE(2);

    // ---- one network pass over the compacted rows
    bool stopped = false;
#define NET_STAGE(n) do { if (stop_after <= (n)) { stopped = true; return 0; } } while (0)
    auto network = [&](int pass) -> int {
        const float* zz = pass == 0 ? b.z_c : b.z_f;
        const int s = pass == 0 ? S : Sf;
        uint32_t* bits = pass == 0 ? b.bits_c : b.bits_f;
        if (pass != 0)       // (pass 0's cull ran in the prologue)
            DANBO_TRY(danbo_bone_cull(bt->rays_o, bt->rays_d, zz, nullptr, R, s, G, bt->skts, m->align, axis_scale, nullptr, nullptr, nullptr, nullptr, bits,
                                      b.row_sample + R, b.cnt,
                                      stream));
        NET_STAGE(21);
        if (pass == 0) DANBO_TRY(join(1));       // the pose volumes and the assignment net's packing
        // ... and side 0 (the trunk's packing, the view constants) HERE, not in front of the trunk where they are needed: round 4's
        // faster pose layers let K2 start while side 0 was still running, and about one step in 200 then had ONE ray's view
        // constants wrong in 16 features (tools/stress_replay.py: 54 of 5 000 replays of one deterministic batch; 0 of 8 000 with
        // this join, 0 of 4 000 with round 3's slow pose layers, which hid it).  Established: inside k_train_cview two evaluations
        // of the same sum from the same LDS rows disagreed in those steps (the last quarter of a wavefront, one of its eight
        // rays) while its LDS rows matched its inputs; K2 beside it is the trigger (not the cull, the pose layers or the packing
        // kernels); the same kernel with its eight FMA chains kept scalar (no v_pk_fma_f32) does not show it under the old order
        // (0 of 6 000 against 9 of 6 000) but is four times slower (71 vs 18 us), so the packed code stays and this join is the
        // fence.  Not established: why (other kernels use packed FMAs beside MFMA all the time, K3 itself does, and are
        // bit-stable).  K2 runs beside nothing in the training step -- which also is 0.7 % faster (1.628 vs 1.640 ms).
        // The order is structural: the product has no switch for the old one (round 4's DANBO_TRAIN_LATE_JOIN exists only in a
        // -DDANBO_DEV_SWITCHES build, for tools/stress_replay.py).
        static const int late_join = dev_env("DANBO_TRAIN_LATE_JOIN", 0);
        if (pass == 0 && !late_join) DANBO_TRY(join(0));
        DANBO_TRY(danbo_gather_assign_blend16_train(bt->rays_o, bt->rays_d, zz, R, s, G, bt->skts, m->align, axis_scale, b.volumes, bits,
                                                    b.row_sample + R, b.cnt, pass == 0 ? nullptr : b.cnt + 1, ncap - R, b.assign16,
                                                    m->p[DANBO_T_A_B0], m->p[DANBO_T_A_B1], m->p[DANBO_T_A_W2], m->p[DANBO_T_A_B2],
                                                    b.h_rows + (size_t)R * 16, b.ticket, stream));
        NET_STAGE(22);
        if (pass == 0 && late_join) DANBO_TRY(join(0));
        // encoding, trunk, heads, raw of the pass (and the row bookkeeping: cnt[1..7], row_ray) in ONE kernel
        return danbo_trunk_fwd(&tw, &trw, pass, stream);
    };
    DANBO_TRY(network(0));
    if (stopped) { DANBO_LAUNCH_RET(); }
    DANBO_STAGE(3);
    if (S <= 64 && Sf <= 64) {
        DANBO_TRY(danbo_composite_importance_fwd(b.raw_c, b.raw_empty, b.bits_c, b.z_c, bt->rays_d, R, S, Sf, B, bt->noise_c, bt->u_rand, o->rgb0,
                                                 o->disp0, o->acc0, b.weights0, o->alpha0, b.z_f, b.z_sorted, b.order, nullptr, nullptr,
                                                 stream));
    } else {
        hipLaunchKernelGGL(k_fill_raw_lazy, dim3(stream_grid((long)R * S, 256)), dim3(256), 0, st, reinterpret_cast<const float4*>(b.raw_empty),
                           b.bits_c, R, S, reinterpret_cast<float4*>(b.raw_c));
        DANBO_TRY(danbo_composite_fwd(b.raw_c, b.z_c, bt->rays_d, R, S, B, bt->noise_c, o->rgb0, o->disp0, o->acc0, b.weights0, o->alpha0, stream));
        DANBO_TRY(danbo_importance_samples(b.z_c, b.weights0, R, S, Sf, bt->u_rand, b.z_f, b.z_sorted, b.order, stream));
    }
    DANBO_STAGE(4);
    DANBO_TRY(network(1));
    if (stopped) { DANBO_LAUNCH_RET(); }
    DANBO_STAGE(5);
    DANBO_TRY(danbo_composite_merged_fwd(b.raw_c, b.raw_f, b.raw_empty, b.bits_c, b.bits_f, b.order, b.z_sorted, bt->rays_d, R, S, Sf, B,
                                         bt->noise_f, o->rgb_map, o->disp_map, o->acc_map, o->weights, o->alpha, b.raw_sorted, nullptr, nullptr,
                                         stream));

    DANBO_STAGE(6);
    // ---- losses and the adjoints of the two composites
    DANBO_TRY(danbo_train_mid(o->rgb_map, o->acc_map, o->rgb0, o->acc0, bt->target, bt->bgs, m->use_background, R, S, Sf, m->loss_mse,
                              m->rgb_loss_coef, m->rgb_loss_coef * m->coarse_weight, B, b.g_rgb, b.g_acc, b.g_rgb0, b.g_acc0, b.raw_c,
                              b.raw_empty, b.raw_sorted, b.bits_c, b.bits_f, b.z_c, b.z_sorted, bt->rays_d, bt->noise_c, bt->noise_f, b.order,
                              o->weights, o->alpha, b.d_raw_c, b.d_raw_f, b.d_raw_rows, b.label_c, b.label_f, b.loss, b.maxabs + MX_RAW,
                              stream));
    DANBO_STAGE(7);
    // per-bone row lists of the K2 adjoint: they need the cull's bits and the forward's row counters only, so they are made HERE, in
    // front of the chain, and the K2 adjoint is the chain's direct successor on the caller's stream (round 5; they used to sit
    // between the two: 9 us on the critical path behind a 15 us cross-queue wait, tools/timeline_train.sh)
    DANBO_TRY(danbo_train_bone_lists(b.bits_c, b.bits_f, b.row_sample, b.cnt, R, ncap, b.lists, b.cntb, stream));
    // ---- the input-gradient chain over the rows of both passes: d raw -> d pre_v -> dz_7 .. dz_0 -> d h
    DANBO_TRY(danbo_trunk_bwd(&tw, &trw, stream));
    DANBO_STAGE(9);
    // ---- three independent branches behind the chain:
    //   side 0: per-ray view gradients (d cview, views_linears.0's per-ray columns, per-camera sums for the frame codes)
    //   side 1: (whole step only; starts behind the K2 adjoint) weight / bias gradients of all dense layers, then -- with side 0's camera sums -- the chain rule of
    //           the merged feature / view layer and the frame codes
    //   main  : K2 / K1b adjoint -> pose GNN adjoint
    ss = (fork_mask & 2) ? ss_all : nullptr;
    if (ss) { s0 = ss->s[0]; s1 = ss->s[1]; } else { s0 = stream; s1 = stream; }
    // Whole step: the view gradients go IN FRONT OF the pose-GNN adjoint on side 1 (they follow the chain, the adjoint follows K2's):
    // two side branches behind the chain ended up on ONE hardware queue in the replayed graph, the view gradients behind the pose
    // adjoint's six launches, and the head chain waited for them (tools/timeline_train.sh, round 5).  Split steps keep side 0.
    static const int dw_on_main = dev_env("DANBO_TRAIN_DW_ON_MAIN", 1);
    const bool vg_on_side1 = ss && phase == 0 && dw_on_main;
    void* vg_stream = vg_on_side1 ? (void*)ss->s[1] : s0;
    if (ss) {
        if (hipEventRecord(ss->fork, st) != hipSuccess) return (int)hipGetLastError();
        if (!vg_on_side1) {
            if (hipStreamWaitEvent(ss->s[0], ss->fork, 0) != hipSuccess) return (int)hipGetLastError();
            guard.pending[0] = true;
        }
        if (phase == 0) {
            if (hipStreamWaitEvent(ss->s[1], ss->fork, 0) != hipSuccess) return (int)hipGetLastError();
            guard.pending[1] = true;
        }
    }
    auto head_chain = [&]() -> int {
        return danbo_train_head_chain(b.g_wfv, b.g_beff, b.csum, m->p[DANBO_T_FEAT_W], m->p[DANBO_T_FEAT_B], m->p[DANBO_T_VIEWS_W],
                                         m->view_ch, m->n_codes, m->code_size, nd, m->g[DANBO_T_FEAT_W], m->g[DANBO_T_FEAT_B],
                                         m->g[DANBO_T_VIEWS_W], m->g[DANBO_T_VIEWS_B], m->n_codes > 0 ? m->g[DANBO_T_CODES] : nullptr, b.vg_part, s1);
    };
    auto pose_adjoint = [&](void* on) -> int {
        return danbo_pose_volumes_bwd(bt->bones, G, m->L_graph, m->graph_width, m->p[DANBO_T_G_W0], m->p[DANBO_T_G_ADJW0], m->g_adj0,
                                      m->p[DANBO_T_G_B0], m->p[DANBO_T_G_W1], m->p[DANBO_T_G_ADJW1], m->g_adj1, m->p[DANBO_T_G_B1],
                                      m->p[DANBO_T_G_W2], m->p[DANBO_T_G_W3], b.vol_scratch, b.g_vol, m->g[DANBO_T_G_W0], m->g[DANBO_T_G_ADJW0],
                                      m->g[DANBO_T_G_B0], m->g[DANBO_T_G_W1], m->g[DANBO_T_G_ADJW1], m->g[DANBO_T_G_B1], m->g[DANBO_T_G_W2],
                                      m->g[DANBO_T_G_B2], m->g[DANBO_T_G_W3], m->g[DANBO_T_G_B3], b.pose_bwd_scratch, on);
    };
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
    // (enqueued BEHIND the K2 adjoint: of the chain's two successors the one captured first continues on the chain's queue, the other
    // starts 15 - 35 us later on another one -- the view gradients have 0.3 ms of slack, the K2 adjoint is the critical path)
    DANBO_TRY(danbo_train_view_grads(b.dpre_v, b.row_ray, b.cnt, ncap, R, b.vin, LD_VIN, m->view_ch, bt->cam_idx, m->n_codes, b.d_cview, b.csum,
                                     m->g[DANBO_T_VIEWS_W], b.vg_part, vg_stream));
    if (ss && phase == 0 && hipEventRecord(ss->join[0], (hipStream_t)vg_stream) != hipSuccess) return (int)hipGetLastError();   // the camera sums
    // The weight-gradient kernel waits for the K2 adjoint: side by side the two just share the compute units (their LDS footprints
    // exclude each other per CU) and the pose-GNN adjoint -- six small launches -- then ran on an otherwise idle device; behind it,
    // those launches hide under the weight gradients (1.71 -> 1.68 ms per step, 0.98 -> 0.94 at 384 rays)
    // Which of the two continues on the caller's stream (round 4, tools/timeline_train.sh): the successor of the K2 adjoint that
    // has to cross to another queue starts 20 - 35 us after it (the cross-queue wait of a replayed graph), the one on its own
    // queue at once.  The weight gradients are the critical path (0.29 ms, the head chain and Adam behind them); the pose-GNN
    // adjoint's six launches have 0.1 ms of slack under them: DANBO_TRAIN_DW_ON_MAIN=1 (default) keeps the former on the
    // caller's stream and sends the latter across.
    void* pose_stream = stream;
    bool pose_done = false;
    // loss terms for the caller: [0] rgb fine, [1] rgb coarse, [2] sum (label - q)^2, [3] volume scale, [4..6] row counters
    auto report_losses = [&](void* on) {
        hipLaunchKernelGGL(k_copy_words_, dim3(1), dim3(64), 0, (hipStream_t)on, reinterpret_cast<const uint32_t*>(b.loss),
                           reinterpret_cast<uint32_t*>(o->loss), 4, reinterpret_cast<const uint32_t*>(b.cnt),
                           reinterpret_cast<uint32_t*>(o->counts), o->counts ? 8 : 0);
        losses_reported = true;
    };
    if (phase == 0 && ss) {
        if (hipEventRecord(ss->mid, st) != hipSuccess || hipStreamWaitEvent(ss->s[1], ss->mid, 0) != hipSuccess) return (int)hipGetLastError();
        if (dw_on_main) {
            pose_stream = s1;
            s1 = stream;
        }
        DANBO_TRY(danbo_dw16(dwl, N_DW, ncap, b.cnt + 4, DW_SLICES, b.dw_scratch, s1));
        if (vg_on_side1) {
            // side 1 carries the view gradients AND the pose adjoint: ONE cross-queue wait in front of the head chain covers the
            // camera sums and ends the fork (round 4 waited for side 0 here and for side 1 behind the head chain: every such wait
            // costs 6 - 12 us of an idle device in the replayed graph although its event completed long ago)
            DANBO_TRY(pose_adjoint(pose_stream));
            pose_done = true;
            report_losses(pose_stream);      // final since the K2 adjoint: off the critical path
            DANBO_TRY(join(1));
        } else if (hipStreamWaitEvent((hipStream_t)s1, ss->join[0], 0) != hipSuccess) {      // recorded behind the view gradients
            return (int)hipGetLastError();
        }
        DANBO_TRY(head_chain());
    }
    DANBO_STAGE(11);
    if (!pose_done) DANBO_TRY(pose_adjoint(pose_stream));
    DANBO_STAGE(12);
    if (phase == 0 && ss) {                        // (the head chain's stream has waited for the view gradients' stream: joining side 1 joins both)
        DANBO_TRY(join(1));
        guard.pending[0] = false;
    } else {
        DANBO_TRY(join(0));
    }
    fused_tail = phase == 0 && ss != nullptr;
    }   // phase != 2
    if (phase == 1) { DANBO_LAUNCH_RET(); }
    // ---- weight / bias gradients of all dense layers (last: it needs nothing but the activations and their gradients, and
    //      data-parallel training hides the all-reduce of everything computed so far -- 7 of the 10 MB -- under it), then the
    //      chain rule of the merged feature / view layer and the frame codes.  (The whole-step call has run both on side 1.)
    if (!fused_tail) {
        DANBO_TRY(danbo_dw16(dwl, N_DW, ncap, b.cnt + 4, DW_SLICES, b.dw_scratch, stream));
        DANBO_TRY(danbo_train_head_chain(b.g_wfv, b.g_beff, b.csum, m->p[DANBO_T_FEAT_W], m->p[DANBO_T_FEAT_B], m->p[DANBO_T_VIEWS_W], m->view_ch,
                                         m->n_codes, m->code_size, nd, m->g[DANBO_T_FEAT_W], m->g[DANBO_T_FEAT_B], m->g[DANBO_T_VIEWS_W],
                                         m->g[DANBO_T_VIEWS_B], m->n_codes > 0 ? m->g[DANBO_T_CODES] : nullptr, b.vg_part, stream));
    }
    // ---- loss terms for the caller: [0] rgb fine, [1] rgb coarse, [2] sum (label - q)^2, [3] volume scale, [4..6] row counters
    if (!losses_reported)
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
