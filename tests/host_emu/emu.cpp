// Host build of the per-item kernel arithmetic (danbo-pytorch_amd/csrc/sample_math.hpp).
// TEST INFRASTRUCTURE: lets the CPU-only test-suite compare the exact code the gfx950
// kernels inline against the numpy oracle.  Compiled with g++ -ffp-contract=off.
#include "../../danbo-pytorch_amd/csrc/sample_math.hpp"

using namespace danbo;

extern "C" {

void emu_coarse_z(const float* nr, const float* fr, int R, int S, float* z) {
    for (int r = 0; r < R; ++r)
        for (int s = 0; s < S; ++s) z[r * S + s] = coarse_z(nr[r], fr[r], s, S);
}

// valid bits + (optionally) pts_t [M,24,3] and part_feat [M,24,15]
void emu_cull_gather(const float* rays_o, const float* rays_d, const float* z, int R, int S, int G, const float* skts,
                     const float* align, const float* axis_scale, const float* volumes, uint32_t* bits, float* pts_t,
                     float* part_feat) {
    float sc[J * 3];
    for (int i = 0; i < J * 3; ++i) sc[i] = fabsf(axis_scale[i]);
    for (long m = 0; m < (long)R * S; ++m) {
        const int r = (int)(m / S), g = r / (R / G);
        float p[3];
        sample_point(rays_o + 3 * r, rays_d + 3 * r, z[m], p);
        uint32_t b = 0;
        for (int j = 0; j < J; ++j) {
            float pt[3];
            bone_local(skts + ((long)g * J + j) * 16, align + j * 16, p, pt);
            if (in_volume(pt, sc + 3 * j)) b |= 1u << j;
            if (pts_t)
                for (int k = 0; k < 3; ++k) pts_t[(m * J + j) * 3 + k] = pt[k];
            if (part_feat) gather_bone_features(volumes + ((long)g * J + j) * VOL, pt, sc + 3 * j, part_feat + (m * J + j) * FEAT);
        }
        bits[m] = b;
    }
}

// returns 1 where the ray hits the cylinder
void emu_cylinder(const float* o, const float* d, const float* cyl, int R, int G, float near0, float far0, float* nr,
                  float* fr, int* hit) {
    for (int r = 0; r < R; ++r) hit[r] = cylinder_bounds(o + 3 * r, d + 3 * r, cyl + 5 * (r / (R / G)), near0, far0, nr + r, fr + r) ? 1 : 0;
}

void emu_boxes(const float* o, const float* d, const float* skts, const float* align, const float* axis_scale, int R,
               int G, float* near_io, float* far_io) {
    float sc[J * 3];
    for (int i = 0; i < J * 3; ++i) sc[i] = fabsf(axis_scale[i]);
    for (int r = 0; r < R; ++r) {
        const int g = r / (R / G);
        float vn = 100000.f, vf = -100000.f;
        bool any = false;
        for (int j = 0; j < J; ++j) {
            float lo, hi;
            if (bone_box_steps(skts + ((long)g * J + j) * 16, align + 16 * j, sc + 3 * j, o + 3 * r, d + 3 * r, &lo, &hi)) {
                any = true;
                vn = fminf(vn, lo);
                vf = fmaxf(vf, hi);
            }
        }
        if (any) { near_io[r] = vn; far_io[r] = vf; }
    }
}

void emu_importance(const float* z, const float* w, int R, int S, int Sf, float* z_fine, float* z_sorted, int32_t* idx) {
    float* cdf = new float[S + 1];
    for (int r = 0; r < R; ++r)
        importance_ray(z + (long)r * S, w + (long)r * S, S, Sf, nullptr, cdf, z_fine + (long)r * Sf,
                       z_sorted + (long)r * (S + Sf), idx + (long)r * (S + Sf));
    delete[] cdf;
}

void emu_composite(const float* raw, const float* z, const float* d, int R, int S, float B, const float* noise,
                   float* rgb, float* disp, float* acc, float* w, float* al) {
    for (int r = 0; r < R; ++r)
        composite_ray(raw + (long)r * S * 4, z + (long)r * S, d + 3 * r, S, B, noise ? noise + (long)r * S : nullptr,
                      rgb + 3 * r, disp + r, acc + r, w + (long)r * S, al + (long)r * S);
}
}
