// danbo_render_frame: the whole eval chain of one ray batch (RayCaster.render_rays, reference core/raycasters.py:245-377, with
// the DANBO network) behind ONE C call -- bounds, coarse samples, per-pose volumes, per-ray view constants, [cull -> gather /
// assignment / blend -> PE + MLP] for the coarse and for the importance samples, composite + resampling (one launch up to 64 + 64
// samples per ray, two beyond), final composite.
// Host code only: it enqueues the kernels of the other translation units on `stream` in the order core/render_engine.py does,
// carving every intermediate out of a caller-provided workspace; no allocation, no synchronisation.
#include "common.hpp"

using namespace danbo;

namespace {
struct Carver {
    char* base;
    size_t used, cap;
    template <class T>
    T* take(size_t n) {
        used = (used + 255) & ~(size_t)255;
        T* p = base ? reinterpret_cast<T*>(base + used) : nullptr;
        used += n * sizeof(T);
        return p;
    }
};

struct FrameBuffers {
    float *near, *far, *cyl_scratch, *z, *vol_scratch, *volumes, *cview, *raw_empty, *h, *raw_a, *raw_b, *z_fine, *z_sorted, *weights0;
    uint32_t *bits_a, *bits_b, *ray_mask, *ray_flat;
    int32_t *list, *count, *order, *ray_list;
};

FrameBuffers carve(Carver& c, int R, int G, int S, int Sf, int chunk, int Wg) {
    FrameBuffers b;
    const size_t M = (size_t)R * S, Mf = (size_t)R * Sf;
    b.near = c.take<float>(R);
    b.far = c.take<float>(R);
    b.cyl_scratch = c.take<float>(8 * (size_t)((R + chunk - 1) / chunk));
    b.z = c.take<float>(M);
    b.vol_scratch = c.take<float>(3 * (size_t)G * 24 * Wg);
    b.volumes = c.take<float>((size_t)G * 24 * 240);
    b.cview = c.take<float>((size_t)R * 128);
    b.raw_empty = c.take<float>((size_t)R * 4);
    b.ray_mask = c.take<uint32_t>(R);
    b.ray_flat = c.take<uint32_t>(R);
    b.ray_list = c.take<int32_t>(R);
    b.bits_a = c.take<uint32_t>(M);
    b.bits_b = c.take<uint32_t>(Mf);
    b.list = c.take<int32_t>(M);          // re-used by the importance pass (Mf <= M is not assumed: max below)
    b.count = c.take<int32_t>(4);         // [0], [1]: rows of the two passes; [2]: K2's tile ticket; [3]: rays that are not rays of constants
    b.h = c.take<float>((M > Mf ? M : Mf) * 16);
    b.raw_a = c.take<float>(M * 4);
    b.raw_b = c.take<float>(Mf * 4);
    b.z_fine = c.take<float>(Mf);
    b.z_sorted = c.take<float>(M + Mf);
    b.order = c.take<int32_t>(M + Mf);
    if (Mf > M) b.list = c.take<int32_t>(Mf);
    b.weights0 = (S <= 64 && Sf <= 64) ? nullptr : c.take<float>(M);     // the coarse weights between the two unfused launches
    return b;
}
}  // namespace

extern "C" size_t danbo_render_frame_workspace(int R, int G, int S, int Sf, int chunk, int graph_width) {
    if (R < 1 || G < 1 || S < 1 || Sf < 1 || chunk < 1 || graph_width < 1) return 0;
    Carver c{nullptr, 0, 0};
    carve(c, R, G, S, Sf, chunk, graph_width);
    return c.used + 256;
}

#define DANBO_TRY(call) do { const int rc_ = (call); if (rc_ != 0) return rc_; } while (0)

extern "C" int danbo_render_frame(const DanboModel* m, const DanboRays* r, int S, int Sf, const DanboFrameOut* o, void* workspace,
                                  size_t workspace_bytes, void* stream) {
    DANBO_CHECK_ARG(m && r && o && workspace && S >= 3 && S <= 256 && Sf >= 1 && Sf <= 64);
    DANBO_CHECK_ARG(r->R >= 1 && r->G >= 1 && r->R % r->G == 0 && r->chunk >= 1);
    DANBO_CHECK_ARG(r->rays_o && r->rays_d && r->skts && r->bones && r->cyls);
    DANBO_CHECK_ARG(o->rgb_map && o->disp_map && o->acc_map && o->alpha && o->weights && o->rgb0 && o->disp0 && o->acc0 && o->alpha0);
    const int R = r->R, G = r->G;
    DANBO_CHECK_ARG(workspace_bytes >= danbo_render_frame_workspace(R, G, S, Sf, r->chunk, m->graph_width));
    Carver c{reinterpret_cast<char*>(((uintptr_t)workspace + 255) & ~(uintptr_t)255), 0, workspace_bytes};
    const FrameBuffers b = carve(c, R, G, S, Sf, r->chunk, m->graph_width);
    hipStream_t st = (hipStream_t)stream;

    // bounds and coarse depths
    DANBO_TRY(danbo_near_far_cylinder(r->rays_o, r->rays_d, r->cyls, R, G, 0.f, 1.f, r->near_in, r->far_in, r->chunk, b.cyl_scratch,
                                      b.near, b.far, stream));
    if (m->use_volume_near_far)
        DANBO_TRY(danbo_near_far_boxes(r->rays_o, r->rays_d, r->skts, m->align, m->axis_scale, R, G, b.near, b.far, stream));
    DANBO_TRY(danbo_coarse_samples(b.near, b.far, R, S, nullptr, b.z, stream));
    // candidate bones of every ray over [near, far]: both culls below skip the rays (and workgroups) that miss every volume
    DANBO_TRY(danbo_ray_bone_mask(r->rays_o, r->rays_d, b.near, b.far, R, G, r->skts, m->align, m->axis_scale, b.ray_mask,
                                  m->flat_rays_ok ? b.ray_flat : nullptr, stream));
    // per pose / per ray
    DANBO_TRY(danbo_pose_volumes_fwd(r->bones, G, m->L_graph, m->graph_width, m->g_w0, m->g_adjw0, m->g_b0, m->g_w1, m->g_adjw1, m->g_b1,
                                     m->g_w2, m->g_b2, m->g_w3, m->g_b3, b.vol_scratch, b.volumes, stream));
    zero_words(b.count, 4, nullptr, 0, st);      // both row counters, the ticket (which returns to 0 after each launch), the ray counter
    const bool flat_rays = m->flat_rays_ok != 0;
    auto view_consts = [&](const int32_t* ray_list, const int32_t* ray_count) -> int {
        return danbo_view_consts(r->rays_d, r->skts, R, G, m->ray_mode, m->normalise, m->L_view, m->framecodes, m->n_codes, m->code_size,
                                 m->mean_code, r->cam_idx, m->views_w_ray_t, m->views_b_eff, m->rgb_w, m->rgb_b, m->empty_consts, 2,
                                 m->code_table, ray_list, ray_count, b.cview, b.raw_empty, stream);
    };
    // the rays of constants (model->flat_rays_ok: flagged by the ray mask; the coarse depths are made here from the same near / far,
    // so they lie inside the interval the flags hold for) get every output of both composites now; the view constants and the
    // composites take the list of the others
    const int32_t *ray_list = nullptr, *ray_count = nullptr;
    if (flat_rays) {
        DANBO_TRY(danbo_flat_rays(b.near, b.ray_flat, R, S, Sf, o->rgb0, o->disp0, o->acc0, b.weights0, o->alpha0, b.z_fine, o->rgb_map,
                                  o->disp_map, o->acc_map, o->weights, o->alpha, b.ray_list, b.count + 3, 3, stream));
        ray_list = b.ray_list;
        ray_count = b.count + 3;
    }
    DANBO_TRY(view_consts(ray_list, ray_count));
    // one network pass over R x s samples at depths zz -> raw (rows outside every volume stay unwritten: bits == 0)
    auto cull = [&](const float* zz, int s, uint32_t* bits, int32_t* count) -> int {
        DANBO_TRY(danbo_bone_cull(r->rays_o, r->rays_d, zz, nullptr, R, s, G, r->skts, m->align, m->axis_scale, b.ray_mask, b.near, b.far,
                                  nullptr, bits, b.list, count, stream));
        // (danbo_group_rows -- rows of the same bone set next to each other -- is not part of the chain any more: it costs what it
        // saves in K2 and scatters K3's rows; core/render_engine.py: group_rows)
        return 0;
    };
    auto network = [&](const float* zz, int s, uint32_t* bits, int32_t* count, float* raw) -> int {
        DANBO_TRY(danbo_gather_assign_blend16_fwd(r->rays_o, r->rays_d, zz, nullptr, R, s, G, r->skts, m->align, m->axis_scale, b.volumes,
                                                  bits, b.list, count, R * s, m->assign16, m->a_b0, m->a_b1, m->a_w2, m->a_b2, b.h,
                                                  nullptr, reinterpret_cast<uint32_t*>(b.count + 2), stream));
        return danbo_pe_mlp32_fwd(b.h, b.list, count, R * s, s, m->mlp16, m->pts_b, m->alpha_w, m->alpha_b, b.cview, m->rgb_w,
                                  m->rgb_b, raw, nullptr, stream);
    };
    DANBO_TRY(cull(b.z, S, b.bits_a, b.count));
    DANBO_TRY(network(b.z, S, b.bits_a, b.count, b.raw_a));
    if (S <= 64) {
        DANBO_TRY(danbo_composite_importance_fwd(b.raw_a, b.raw_empty, b.bits_a, b.z, r->rays_d, R, S, Sf, m->density_scale, nullptr,
                                                 nullptr, o->rgb0, o->disp0, o->acc0, nullptr, o->alpha0, b.z_fine, b.z_sorted, b.order,
                                                 ray_list, ray_count, stream));
    } else {     // rays of more than 64 coarse samples: the same two steps as two launches, the weights in between
        DANBO_TRY(danbo_composite_rays_fwd(b.raw_a, b.raw_empty, b.bits_a, b.z, r->rays_d, R, S, m->density_scale, nullptr, o->rgb0,
                                           o->disp0, o->acc0, b.weights0, o->alpha0, ray_list, ray_count, stream));
        DANBO_TRY(danbo_importance_samples_rays(b.z, b.weights0, R, S, Sf, nullptr, b.z_fine, b.z_sorted, b.order, ray_list, ray_count,
                                                stream));
    }
    DANBO_TRY(cull(b.z_fine, Sf, b.bits_b, b.count + 1));
    DANBO_TRY(network(b.z_fine, Sf, b.bits_b, b.count + 1, b.raw_b));
    return danbo_composite_merged_fwd(b.raw_a, b.raw_b, b.raw_empty, b.bits_a, b.bits_b, b.order, b.z_sorted, r->rays_d, R, S, Sf,
                                      m->density_scale, nullptr, o->rgb_map, o->disp_map, o->acc_map, o->weights, o->alpha, nullptr,
                                      ray_list, ray_count, stream);
}
