/*
 * libdanbo_hip.so -- C ABI of the MI355X (gfx950) articulated-NeRF rendering path.
 *
 * Drop-in boundary: the Python classes in danbo-pytorch_amd/core/networks
 * (same names / state_dict keys as the reference's core/networks) call these entry points
 * through ctypes.  Every pointer is a DEVICE pointer to row-major float32 unless noted; no
 * function retains a pointer past the call; every kernel is enqueued on `stream` (a
 * hipStream_t passed as void*) and nothing synchronises with the host.  Return value:
 * 0 = ok, otherwise a hipError_t, or DANBO_EINVAL (-22) for a rejected argument.
 *
 * Each entry cites the reference code (file:line under the reference tree) it replaces.
 * J = 24 bones everywhere.  "pose" g of ray r is r / (R / G) (equal contiguous groups,
 * core/encoders.py:465-468, core/networks/gnn_backbone.py:792-794).
 */
#ifndef DANBO_HIP_H
#define DANBO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DANBO_EINVAL (-22)
#define DANBO_J 24
#define DANBO_VOL 240       /* voxel_feat(5) * voxel_res(16) * 3 axes               */
#define DANBO_FEAT 15       /* voxel_feat * 3 ('cat' construct)                     */
#define DANBO_H_STRIDE 16   /* blended feature rows are padded 15 -> 16 floats      */
#define DANBO_RAY_FLAT_VMAX 1e4f /* danbo_ray_bone_mask flags no ray whose view direction inputs may exceed this */

/* library / device identification (host only).
 * ABI history: 2 = danbo_adam_step takes the step's scalars by value; 3 = DanboAssignBwd.d_p; 4 = danbo_train_workspace_view,
 * danbo_group_rows (additive); 5 = danbo_ray_bone_mask, danbo_flat_rays;
 * danbo_bone_cull takes the mask and the flags, the two fused composites a ray list (all nullable); additive since:
 * danbo_composite_rays_fwd, danbo_importance_samples_rays, danbo_random_draws, danbo_gather_rows; danbo_render_frame takes up to
 * 256 + 64 samples per ray; 6 = DANBO_MLP16_PACKED_BYTES grows by a trailer (power-of-two pack scales of danbo_mlp16_pack);
 * 7 = danbo_train_mid (additive); DanboTrainBatch.rng_* (appended: a caller of an older header must be rebuilt);
 * 8 = A-NeRF on the library's own kernels end to end (additive): danbo_anerf_view_consts_fwd / _bwd, danbo_anerf_train_step and its
 * building blocks; 9 = K3 in the 32x32x16 form (danbo_mlp32_pack, danbo_pe_mlp32_fwd: additive), danbo_view_consts' rgb_order 2;
 * danbo_render_frame runs it: DanboModel.mlp16 is a buffer packed by danbo_mlp32_pack (a caller of ABI 8 must re-pack); additive:
 * danbo_transform_batch_pts, danbo_optcodes_fwd (the reference's eager encoder helpers). */
int danbo_abi_version(void);
int danbo_device_info(int* cu_count, int* lds_bytes, char* arch, int arch_len);

/* ---------------------------------------------------------------------------------------
 * Pose stage (once per distinct pose): axis-angle -> rot6d -> PE(L) -> skeleton GNN ->
 * factorised volumes.   Replaces SamplePointsEmbedder.encode_graph_inputs
 * (core/encoders.py:460-473), AxisAngtoRot6DEncoder (:859-877), Embedder
 * (core/cutoff_embedder.py:62-73) and FactorizeGNN/BodyGNN.forward
 * (core/networks/gnn_backbone.py:683-704) for gcn_D=4, gcn_fc_D=1:
 *   GCN(6(1+2L)->W) , GCN(W->W) , PerBoneLinear(W->W) , PerBoneLinear(W->240)
 * incl. mask_root and the doubled first layer (skip_gcn=False quirk).
 *   bones [G,24,3]; w0 [24,Cin,W]; adjw0/adjw1 [24,24] (already adj_w*adj); b0,b1 [W];
 *   w1,w2 [24,W,W]; b2 [24,W]; w3 [24,W,240]; b3 [24,240]; scratch >= 3*G*24*W floats.
 * ------------------------------------------------------------------------------------- */
int danbo_pose_volumes_fwd(const float* bones, int G, int L_graph, int W,
                           const float* w0, const float* adjw0, const float* b0,
                           const float* w1, const float* adjw1, const float* b1,
                           const float* w2, const float* b2,
                           const float* w3, const float* b3,
                           float* scratch, float* volumes /*[G,24,240]*/, void* stream);

/* ---------------------------------------------------------------------------------------
 * Ray bounds.  get_near_far_in_cylinder (core/utils/ray_utils.py:294-346) incl. the
 * per-`chunk` nan-mean back-fill (the reference is invoked once per `chunk` rays,
 * core/trainer.py:75-90) done on device -- no host round trip.
 *   rays_o, rays_d [R,3]; cyl [G,5]; near0/far0: the placeholder bounds (0/1,
 *   core/trainer.py:96-98); scratch >= 32*ceil(R/chunk) BYTES (fp64 partial sums).
 * ------------------------------------------------------------------------------------- */
int danbo_near_far_cylinder(const float* rays_o, const float* rays_d, const float* cyl, int R, int G,
                            float near0, float far0,
                            const float* near_in /*[R] or NULL: per-ray placeholders override near0*/,
                            const float* far_in /*[R] or NULL*/, int chunk, float* scratch,
                            float* near_out /*[R]*/, float* far_out /*[R]*/, void* stream);

/* GraphCaster.get_near_far + get_ray_box_intersections (core/raycasters.py:648-707,
 * core/utils/ray_utils.py:383-417), fp64 plane hits, bound fixed at 1.3; updates near/far
 * in place for rays that hit >= 1 bone box.   skts [G,24,4,4]; align [24,4,4];
 * axis_scale [24,3]. */
int danbo_near_far_boxes(const float* rays_o, const float* rays_d, const float* skts, const float* align,
                         const float* axis_scale, int R, int G, float* near_io, float* far_io, void* stream);

/* sample_from_lineseg, perturb=0 / stratified with caller-supplied uniforms
 * (core/utils/ray_utils.py:206-253): z [R,S]; t_rand NULL or [R,S]. */
int danbo_coarse_samples(const float* near, const float* far, int R, int S, const float* t_rand,
                         float* z, void* stream);

/* ---------------------------------------------------------------------------------------
 * K1a  world->bone transform + in-volume cull over ALL samples of the pass.
 * transform_batch_pts + align (core/encoders.py:288-303,442-444) and the `invalid` test of
 * FactorizeGNN.sample_from_volume (core/networks/gnn_backbone.py:802,808), bit-exact with
 * oracle/danbo_oracle.py (unfused fp32 mul/add chain).
 *   pts = rays_o + rays_d * z  (core/raycasters.py:463), or read from `pts` [R*S,3] when the
 *   caller already holds sample points (the reference's model(inputs['pts']) entry); exactly
 *   one of {z, pts} is non-NULL.
 *   valid_bits[m] bit j = sample m inside bone j's volume.
 *   If `list` != NULL: indices of samples with valid_bits != 0 are appended to list and
 *   *count (which the caller zeroes) is incremented -- order unspecified.
 *   ray_mask / t_lo / t_hi (all three or none; z mode only): the result of danbo_ray_bone_mask for
 *   these rays.  Purely an accelerator -- the per-sample test is the same and a depth outside
 *   [t_lo, t_hi] of its ray is tested against every bone, so valid_bits never depends on it;
 *   workgroups whose rays miss every volume write their zeros and leave (a 512 x 512 x 48 pass:
 *   133 -> 73 us).  Without it the kernel derives the same rejection per 1024-sample window.
 *   ray_flat (optional, with ray_mask): danbo_ray_bone_mask's flags; cleared for every ray with a
 *   depth outside its interval, so that a flag still set afterwards means "no sample of this pass,
 *   nor any depth in between, is inside a volume" (see danbo_flat_rays).
 * ------------------------------------------------------------------------------------- */
int danbo_bone_cull(const float* rays_o, const float* rays_d, const float* z, const float* pts, int R, int S, int G,
                    const float* skts /*[G,24,4,4]*/, const float* align /*[24,4,4]*/,
                    const float* axis_scale /*[24,3]*/, const uint32_t* ray_mask /*[R] or NULL*/,
                    const float* t_lo /*[R] or NULL*/, const float* t_hi /*[R] or NULL*/,
                    uint32_t* ray_flat /*[R] or NULL, in/out*/,
                    uint32_t* valid_bits /*[R*S]*/, int32_t* list /*[R*S] or NULL*/,
                    int32_t* count /*[1] or NULL*/, void* stream);

/* Per-ray candidate bones: ray_mask[r] bit j = 0 when no point rays_o + t rays_d with
 * t_lo[r] <= t <= t_hi[r] can lie inside the volume of bone j (slab test of the segment against the
 * slightly inflated box, conservative).  Once per ray batch -- the coarse and the importance samples
 * of a ray both lie inside its [near, far].  ray_flat (optional) [R]: 1 when ray_mask[r] == 0, every
 * interval length the ray can produce inside [t_lo, t_hi] is finite, and |rays_d| times max(1, the
 * absolute row sums of the root bone's 3x3 matrix) -- a bound on every view-direction input of the
 * colour branch -- is at most DANBO_RAY_FLAT_VMAX: the ray is a candidate for the constants of an empty
 * ray (danbo_flat_rays).  No reference counterpart: the reference evaluates every sample against
 * every bone (core/networks/gnn_backbone.py:787-828); this feeds danbo_bone_cull and the composites. */
int danbo_ray_bone_mask(const float* rays_o, const float* rays_d, const float* t_lo /*[R]*/, const float* t_hi /*[R]*/,
                        int R, int G, const float* skts, const float* align, const float* axis_scale,
                        uint32_t* ray_mask /*[R]*/, uint32_t* ray_flat /*[R] or NULL*/, void* stream);

/* K1b  factorised tri-axis gather (factorize_grid_sample, core/networks/misc.py:331-351;
 * windowing + 'cat' construct, gnn_backbone.py:803-826) for the n listed samples
 * (list == NULL: samples 0..n-1; count != NULL: n is read from *count on device and the
 * host `n` is only the capacity).  part_feat [n,24,15] is the reference's stage-boundary
 * tensor (1440 B/sample); this is the HBM-bound kernel of the path. */
int danbo_bone_gather_fwd(const float* rays_o, const float* rays_d, const float* z, const float* pts, int R, int S, int G,
                          const float* skts, const float* align, const float* axis_scale,
                          const float* volumes /*[G,24,240]*/,
                          const int32_t* list, const int32_t* count, int n,
                          float* part_feat /*[n,24,15]*/, void* stream);

/* K2  per-sample bone-assignment GNN + masked sigmoid + blend
 * (MixGNN, gnn_backbone.py:567-591,602-629; DANBO.sigmoid / blend, danbo.py:299-300,406-415)
 *   w0 [24,15,32]; adjw [24,24]; b0 [32]; w1 [24,32,32]; b1 [24,32]; w2 [24,32]; b2 [24]
 *   valid_bits indexed by sample id (list[i] or i); outputs: h [n,16] (15 + zero pad),
 *   confd [n,24] or NULL. */
int danbo_assign_blend_fwd(const float* part_feat, const uint32_t* valid_bits,
                           const int32_t* list, const int32_t* count, int n,
                           const float* w0, const float* adjw, const float* b0,
                           const float* w1, const float* b1, const float* w2, const float* b2,
                           float* h, float* confd, void* stream);

/* K1b + K2 fused (render path): every bone lane recomputes its own transform + gather, the
 * 1440 B/sample part_feat tensor is never written.  Same results as K1b followed by K2. */
int danbo_gather_assign_blend_fwd(const float* rays_o, const float* rays_d, const float* z, const float* pts,
                                  int R, int S, int G, const float* skts, const float* align,
                                  const float* axis_scale, const float* volumes,
                                  const uint32_t* valid_bits, const int32_t* list, const int32_t* count, int n,
                                  const float* w0, const float* adjw, const float* b0,
                                  const float* w1, const float* b1, const float* w2, const float* b2,
                                  float* h, float* confd, void* stream);

/* In-place re-ordering of the compacted in-volume rows (list[0 .. *count)) so that rows whose samples lie inside the SAME SET of
 * bone volumes (equal valid_bits words) are neighbours, inside windows of 16 384 rows (csrc/k_group.hip).  No reference
 * counterpart: the reference evaluates every sample against every bone (core/networks/gnn_backbone.py:787-828); this serves the
 * launch that follows -- danbo_gather_assign_blend16_fwd evaluates, per wavefront of 32 consecutive rows, every bone valid for
 * at least one of them.  The order of the list is free: every consumer (the h rows, K3's scatter) goes through it and a row's
 * result does not depend on its neighbours.  count: device scalar or NULL (= n_cap rows). */
int danbo_group_rows(const uint32_t* valid_bits, int32_t* list, const int32_t* count, int n_cap, void* stream);
/* dev tool (tools/micro_assign.py --trace): int64[256] device buffer for (tag, s_memtime) stamps of one wavefront of
 * danbo_gather_assign_blend16_fwd, NULL = off */
int danbo_assign16_set_trace(void* buf);

/* K1b + K2, fast variant (csrc/k_assign16.hip): same contract as danbo_gather_assign_blend_fwd with the
 * two per-bone GEMMs on fp16 hi/lo-split MFMAs (fp32 accumulate) and the skeleton adjacency folded
 * into the layer-0 weights by danbo_assign16_pack (w0 [24,15,32], adjw [24,24], w1 [24,32,32]). */
#define DANBO_ASSIGN16_PACKED_BYTES 262144
int danbo_assign16_pack(const float* w0, const float* adjw, const float* w1, void* packed16, void* stream);
int danbo_gather_assign_blend16_fwd(const float* rays_o, const float* rays_d, const float* z, const float* pts,
                                    int R, int S, int G, const float* skts, const float* align,
                                    const float* axis_scale, const float* volumes,
                                    const uint32_t* valid_bits, const int32_t* list, const int32_t* count, int n,
                                    const void* packed16, const float* b0, const float* b1 /*[24,32]*/,
                                    const float* w2 /*[24,32]*/, const float* b2 /*[24]*/,
                                    float* h, float* confd, uint32_t* ticket, void* stream);
/* ticket (ABI 4): one device word the launch hands its 128-row tiles out with -- a tile costs 1 .. 40 x the cheapest one,
 * depending on how many bone volumes its rows lie in.  It must be 0 at launch and is 0 again when the launch has finished (the
 * counter wraps on the last draw), so one word allocated and zeroed ONCE serves every launch enqueued on the same stream. */

/* ---------------------------------------------------------------------------------------
 * K3  voxel-feature PE + density/colour MLP on fp32 MFMA (v_mfma_f32_32x32x2_f32).
 * Embedder (core/cutoff_embedder.py:62-73), NeRF.inference / forward_density /
 * forward_view (core/networks/nerf.py:176-209) for D=8, W=256, skip after layer 4,
 * view_W=128.  The per-ray part of the view layer (PE(dir) | frame code) is hoisted into
 * `cview` by danbo_view_consts.
 * Weights must first be repacked into MFMA fragment order with danbo_mlp_pack.
 * ------------------------------------------------------------------------------------- */
#define DANBO_MLP_PACKED_FLOATS 659456  /* see csrc/k_mlp.hip */
int danbo_mlp_pack(const float* const* pts_w /*8 ptrs [256,195|256|451]*/, const float* feature_w /*[256,256]*/,
                   const float* views_w /*[128,256+Cv]*/, int Cv,
                   float* packed /*[DANBO_MLP_PACKED_FLOATS]*/, float* views_w_ray_t /*[Cv,128]*/, void* stream);

/* per-ray constants: view direction transform (core/encoders.py:179-189,570-578,774-795),
 * PE(L_view), frame code lookup (core/networks/embedding.py:17-39; cam_idx NULL or <0 ->
 * mean code), cview[r] = W_view[:,256:] . [PE(dir)|code] + b_view;  if empty_consts != NULL
 * also raw_empty[r] = (rgb_linear(relu(empty_view_pre + cview[r])), empty_alpha): the raw
 * of every sample of ray r that lies in no bone volume (blended feature h = 0).
 *   ray_mode 0: 'world' raw rays_d; 1: 'root_local' (skts[g,0,:3,:3] . d);  normalise: relray
 *   ray_list / ray_count (optional, together): only the listed rays' rows are computed and written
 *   (danbo_flat_rays' list: nobody reads the rows of a ray of constants).
 *   views_b, empty_consts and code_table are read 16 bytes at a time: each must be 16-byte aligned (DANBO_EINVAL otherwise). */
int danbo_view_consts(const float* rays_d, const float* skts, int R, int G, int ray_mode, int normalise,
                      int L_view, const float* framecodes /*[n_codes,Cf] or NULL*/, int n_codes, int Cf,
                      const float* mean_code /*[Cf]: codes.mean(0), used where cam_idx < 0*/,
                      const int64_t* cam_idx /*[R] or NULL*/,
                      const float* views_w_ray_t /*[Cv,128]*/, const float* views_b /*[128]*/,
                      const float* rgb_w /*[3,128]*/, const float* rgb_b /*[3]*/,
                      const float* empty_consts /*[129] or NULL*/,
                      int rgb_order /*0: summation order of danbo_pe_mlp_fwd, 1: of danbo_pe_mlp16_fwd, 2: of danbo_pe_mlp32_fwd*/,
                      const float* code_table /*[n_codes+1,128] from danbo_view_code_table, or NULL*/,
                      const int32_t* ray_list /*[R] or NULL*/, const int32_t* ray_count /*[1] or NULL*/,
                      float* cview /*[R,128]*/, float* raw_empty /*[R,4] or NULL*/, void* stream);

/* Per-camera part of the view constants, once per weight update: table[c] = views_b + W_view[:, code
 * columns] . framecode[c] for c < n_codes, row n_codes = the mean code.  With it danbo_view_consts
 * only sums the 3(1+2L) direction encodings per ray. */
int danbo_view_code_table(const float* framecodes, const float* mean_code, int n_codes, int Cf, int L_view,
                          const float* views_w_ray_t, const float* views_b, float* table /*[n_codes+1,128]*/,
                          void* stream);

/* rows: h [n,16] (output of K2).  Row i belongs to sample id m = list ? list[i] : i and ray
 * m / S; raw_out[m] = (rgb logits, density logit).  aux_out (optional) [n,129] receives the
 * view-layer pre-activation WITHOUT cview and the density logit (used once per weight
 * update to derive empty_consts from a zero row). */
int danbo_pe_mlp_fwd(const float* h, const int32_t* list, const int32_t* count, int n, int S,
                     const float* packed, const float* const* pts_b /*8 ptrs [256]*/,
                     const float* alpha_w /*[256]*/, const float* alpha_b /*[1]*/,
                     const float* feature_b /*[256]*/, const float* cview /*[R,128] or NULL*/,
                     const float* rgb_w, const float* rgb_b,
                     float* raw_out /*[R*S,4]*/, float* aux_out, void* stream);

/* K3, fast variant (csrc/k_mlp16.hip): same contract as danbo_pe_mlp_fwd, products evaluated as
 * three fp16 MFMAs on hi/lo-split operands with fp32 accumulation (<= 2e-6 relative on the raw
 * logits vs fp64 -- the accuracy class of an fp32 GEMM).  danbo_mlp16_pack writes
 * DANBO_MLP16_PACKED_BYTES bytes of fp16 fragments; it also merges feature_linear with the
 * per-sample part of views_linears.0 (no activation in between, nerf.py:200-204) into one
 * 256->128 GEMM and returns the matching view bias views_b_eff = views_b + W_v[:, :256] feature_b,
 * which the caller passes to danbo_view_consts in place of views_b.
 * ABI 6: every matrix is packed times the power of two that puts its largest |entry| into [2^13, 2^14) (unscaled, the lo half of
 * every weight below 2^-3 is an fp16 subnormal); the exact inverses travel in the buffer's trailer and the kernel applies them
 * inside the fma that adds the bias -- the caller sees nothing but the larger DANBO_MLP16_PACKED_BYTES. */
#define DANBO_MLP16_TRAILER_BYTES (128 + 128 * 256 * 4)   /* winv [16] | wmax [16] | W_fv [128,256]: written by danbo_mlp16_pack */
#define DANBO_MLP16_PACKED_BYTES (2424832 + DANBO_MLP16_TRAILER_BYTES)
int danbo_mlp16_pack(const float* const* pts_w, const float* feature_w, const float* feature_b,
                     const float* views_w, const float* views_b, int Cv,
                     void* packed16, float* views_b_eff /*[128]*/, void* stream);
int danbo_pe_mlp16_fwd(const float* h, const int32_t* list, const int32_t* count, int n, int S,
                       const void* packed16, const float* const* pts_b,
                       const float* alpha_w, const float* alpha_b,
                       const float* cview, const float* rgb_w, const float* rgb_b,
                       float* raw_out, float* aux_out, void* stream);

/* K3, 32x32x16 form (csrc/k_mlp32.hip, ABI 9): the contract, arithmetic and buffer size of the pair above on
 * v_mfma_f32_32x32x16_f16 -- one wavefront of 32 samples per SIMD with both result banks in AccVGPRs, half the LDS fragment reads per
 * flop, the k-substep's epilogue hand-interleaved into the MFMA stream (9.5 % faster per launch on the bench frame).  The fragment order
 * differs: a buffer packed by danbo_mlp32_pack is for danbo_pe_mlp32_fwd only.  Results differ from danbo_pe_mlp16_fwd in the last
 * bits (the products of a k-step are added in another order: 7e-7 of the channel range on the bench frame); a caller that needs
 * the per-ray empty-space raw of danbo_view_consts bit-equal to this kernel's passes rgb_order = 2 there. */
int danbo_mlp32_pack(const float* const* pts_w, const float* feature_w, const float* feature_b,
                     const float* views_w, const float* views_b, int Cv,
                     void* packed32 /*DANBO_MLP16_PACKED_BYTES*/, float* views_b_eff /*[128]*/, void* stream);
int danbo_pe_mlp32_fwd(const float* h, const int32_t* list, const int32_t* count, int n, int S,
                       const void* packed32, const float* const* pts_b,
                       const float* alpha_w, const float* alpha_b,
                       const float* cview, const float* rgb_w, const float* rgb_b,
                       float* raw_out, float* aux_out, void* stream);

/* The reference's eager encoder helpers (csrc/k_encoders.hip; additive in ABI 9) for callers that use them outside the fused path:
 *   danbo_transform_batch_pts  core/encoders.py:288-303 transform_batch_pts: out[r, s, j, :] = skt[r / rays_per_pose, j] applied to the
 *                              point pts[r, s, :] (rot_only = 1: core/encoders.py:305-318 transform_batch_rays, the 3 x 3 part applied to
 *                              a direction); pts [n_rays, S, 3], skt [n_rays / rays_per_pose, J, 4, 4], out [n_rays, S, J, 3]
 *   danbo_optcodes_fwd         core/networks/embedding.py:17-39 Optcodes.forward on codes [n_codes, code_ch]: mode 0 row lookup (idx
 *                              [N, idx_cols] as floats, column 0; clamped to the table), 1 the mean code for every row (evaluation
 *                              without a frame: idx < 0), 2 torch.lerp of the rows idx[:, 0], idx[:, 1] with weight idx[:, 2] */
int danbo_transform_batch_pts(const float* pts, const float* skt, long n_rays, int S, int J, int rays_per_pose, int rot_only,
                              float* out, void* stream);
int danbo_optcodes_fwd(const float* codes, int n_codes, int code_ch, const float* idx, int idx_cols, long N, int mode, float* out,
                       void* stream);

/* raw[r,s,:] = raw_empty[r,:] (broadcast fill before K3 scatters the in-volume rows) */
int danbo_fill_raw(const float* raw_empty, int R, int S, float* raw, void* stream);

/* ---------------------------------------------------------------------------------------
 * K4  NeRF.raw2outputs (core/networks/nerf.py:281-347), relu density, one wavefront per ray.
 *   raw [R,S,4]; z [R,S]; rays_d [R,3]; noise NULL or [R,S] (already scaled by
 *   raw_noise_std*B); outputs rgb_map [R,3], disp [R], acc [R], weights [R,S], alpha [R,S].
 * ------------------------------------------------------------------------------------- */
int danbo_composite_fwd(const float* raw, const float* z, const float* rays_d, int R, int S, float B,
                        const float* noise, float* rgb_map, float* disp, float* acc,
                        float* weights, float* alpha, void* stream);
/* the same with danbo_composite_importance_fwd's two conventions, for rays too long for that call (S > 64):
 * valid_bits / raw_empty (optional, together): samples whose in-volume word is 0 were not written to raw and take raw_empty[ray];
 * ray_list / ray_count (optional, together): only the listed rays (danbo_flat_rays' list; *ray_count is read on the device) are
 * composited, every output row of an unlisted ray is left untouched.  All four NULL = danbo_composite_fwd. */
int danbo_composite_rays_fwd(const float* raw, const float* raw_empty /*[R,4]*/, const uint32_t* valid_bits /*[R,S]*/, const float* z,
                             const float* rays_d, int R, int S, float B, const float* noise, float* rgb_map, float* disp,
                             float* acc, float* weights, float* alpha, const int32_t* ray_list, const int32_t* ray_count,
                             void* stream);

/* K4 backward: gradients of rgb_map [R,3] and acc_map [R] -> d raw [R,S,4] (S <= 256).  disp_map,
 * weights and alpha carry no gradient in the reference's losses (core/trainer.py:396-422,507-536). */
int danbo_composite_bwd(const float* raw, const float* z, const float* rays_d, int R, int S, float B,
                        const float* noise, const float* g_rgb, const float* g_acc, float* d_raw, void* stream);
/* the same on a lazily filled raw tensor (danbo_composite_importance_fwd's convention): samples whose in-volume word is 0
 * take raw_empty[ray] */
int danbo_composite_bwd_lazy(const float* raw, const float* raw_empty /*[R,4]*/, const uint32_t* valid_bits /*[R,S] or NULL*/,
                             const float* z, const float* rays_d, int R, int S, float B, const float* noise,
                             const float* g_rgb, const float* g_acc, float* d_raw, void* stream);

/* K1b backward: d part_feat [n,24,15] -> d volumes [G,24,240] and d axis_scale [24,3] (both ACCUMULATED
 * with atomics: the caller zeroes them).  Same autograd semantics as the reference: the window is
 * detached and the in-volume mask is not differentiable (gnn_backbone.py:802-808). */
int danbo_bone_gather_bwd(const float* rays_o, const float* rays_d, const float* z, const float* pts,
                          int R, int S, int G, const float* skts, const float* align, const float* axis_scale,
                          const float* volumes, const int32_t* list, int n, const float* d_part_feat,
                          float* d_volumes, float* d_axis_scale, void* stream);

/* isample_from_lineseg(is_only=True) + sample_pdf (core/utils/ray_utils.py:159-203,257-291)
 * + the sort of [z, z_fine] (stable, coarse first):  u NULL = linspace(0,1,Sf) (det) else
 * caller-supplied uniforms [R,Sf].  sorted_idx int32 [R,S+Sf]. */
int danbo_importance_samples(const float* z, const float* weights, int R, int S, int Sf, const float* u,
                             float* z_fine /*[R,Sf]*/, float* z_sorted /*[R,S+Sf]*/,
                             int32_t* sorted_idx /*[R,S+Sf]*/, void* stream);
/* the same over a list of rays (64 < S <= 256 and Sf <= 64 with a list; both NULL = danbo_importance_samples): the rows of
 * unlisted rays are left untouched (danbo_flat_rays wrote their z_fine; nobody reads their z_sorted / sorted_idx) */
int danbo_importance_samples_rays(const float* z, const float* weights, int R, int S, int Sf, const float* u,
                                  float* z_fine, float* z_sorted, int32_t* sorted_idx, const int32_t* ray_list,
                                  const int32_t* ray_count, void* stream);

/* merge_samples (core/raycasters.py:745-761): out[r,i,:] = cat(a[r],b[r])[sorted_idx[r,i],:] */
int danbo_merge_samples(const float* a /*[R,S,C]*/, const float* b /*[R,Sf,C]*/, const int32_t* sorted_idx,
                        int R, int S, int Sf, int C, float* out /*[R,S+Sf,C]*/, void* stream);

/* raw2outputs of the coarse pass + isample_from_lineseg in one launch (S, Sf <= 64; same arithmetic as
 * danbo_composite_fwd followed by danbo_importance_samples).  valid_bits / raw_empty (optional, together):
 * samples whose in-volume word is 0 were not written to raw and take raw_empty[ray] instead.
 * weights / alpha may be NULL.
 * ray_list / ray_count (optional, together): composite only the listed rays (danbo_flat_rays' list; *ray_count is read on the
 * device) -- every output row of an unlisted ray is left untouched. */
int danbo_composite_importance_fwd(const float* raw /*[R,S,4]*/, const float* raw_empty /*[R,4]*/,
                                   const uint32_t* valid_bits /*[R,S]*/, const float* z, const float* rays_d, int R, int S,
                                   int Sf, float B, const float* noise, const float* u, float* rgb_map, float* disp,
                                   float* acc, float* weights, float* alpha, float* z_fine, float* z_sorted,
                                   int32_t* sorted_idx, const int32_t* ray_list /*[R] or NULL*/,
                                   const int32_t* ray_count /*[1] or NULL*/, void* stream);

/* Rays of constants (no reference counterpart; the values are the reference's).  ray_flat: the flags danbo_ray_bone_mask made
 * and danbo_bone_cull of the COARSE pass (same rays, depths, mask, interval) has passed on -- a set flag says the ray cannot
 * meet a volume anywhere in [t_lo, t_hi], which holds all of its coarse depths: every sample of the ray, in the coarse AND in the
 * importance pass (whose depths lie between the coarse ones), carries the ray's empty-space raw.  The CALLER states that
 *   (a) the model's empty-space density empty_consts[128] / B is <= 0, and
 *   (b) the empty-space colour logits cannot be NaN for the flagged rays: a finite bound from the weights' magnitudes with
 *       view-direction inputs of at most DANBO_RAY_FLAT_VMAX (core/render_engine.py:_flat_rays_ok),
 * both properties of the weights; then sig = 0, alpha = +0, T = 1, w = +0 on every sample and raw2outputs gives +0 maps for both
 * passes.  This call writes exactly those outputs for the flagged rays -- the coarse composite's (rgb0, disp0, acc0, weights0 /
 * alpha0 rows, may be NULL) and the merged composite's (rgb_map, disp, acc, weights / alpha rows, may be NULL) -- sets their z_fine
 * row to t_lo (so that the importance pass's danbo_bone_cull drops them on their mask; their z_sorted / sorted_idx rows are never
 * made), and appends every OTHER ray to ray_list / *ray_count (zeroed by the caller), to be passed to danbo_view_consts,
 * danbo_composite_importance_fwd and danbo_composite_merged_fwd (any S, Sf; rays of more than 64 coarse samples:
 * danbo_composite_rays_fwd, danbo_importance_samples_rays, danbo_composite_merged_fwd).  All maps / alphas / weights of the frame are then bit-identical
 * to evaluating every ray; a caller that wants z_fine / z_sorted / sorted_idx / cview / raw_empty of every ray, density noise, or
 * cannot state (a) and (b), must not use it.
 * parts: 1 = the list and the per-ray outputs (a few us: what danbo_view_consts waits for), 2 = the per-sample rows (weights0 /
 * alpha0 / z_fine / weights / alpha: ~130 MB of a 512 x 512 frame, needed only by the composites' consumers), 3 = both. */
int danbo_flat_rays(const float* t_lo /*[R]*/, const uint32_t* ray_flat /*[R]*/, int R, int S, int Sf, float* rgb0, float* disp0,
                    float* acc0, float* weights0, float* alpha0, float* z_fine, float* rgb_map, float* disp, float* acc,
                    float* weights /*[R,S+Sf]*/, float* alpha, int32_t* ray_list /*[R]*/, int32_t* ray_count /*[1]*/, int parts,
                    void* stream);

/* merge_samples (core/raycasters.py:745-761) folded into raw2outputs of the merged samples: sample i of the sorted
 * order is read from raw_a (sorted_idx < S) or raw_b; raw_sorted (optional) receives the merged raw tensor. */
int danbo_composite_merged_fwd(const float* raw_a /*[R,S,4]*/, const float* raw_b /*[R,Sf,4]*/, const float* raw_empty,
                               const uint32_t* bits_a, const uint32_t* bits_b, const int32_t* sorted_idx,
                               const float* z_sorted /*[R,S+Sf]*/, const float* rays_d, int R, int S, int Sf, float B,
                               const float* noise, float* rgb_map, float* disp, float* acc, float* weights /*[R,S+Sf]*/,
                               float* alpha, float* raw_sorted /*[R,S+Sf,4] or NULL*/,
                               const int32_t* ray_list /*[R] or NULL: as danbo_composite_importance_fwd*/,
                               const int32_t* ray_count, void* stream);

/* ---------------------------------------------------------------------------------------------
 * A-NeRF (nerf_type = nerf): the per-sample encoders around the W = 448 trunk.
 * The trunk's dense layers are plain GEMMs on the rows produced here (core/anerf_engine.py).
 * ------------------------------------------------------------------------------------------- */

/* SamplePointsEmbedder.encode_pts (core/encoders.py:424-450) with RelDistEncoder (:630-651) and
 * VecNormEncoder (:774-795), then CutoffEmbedder._embed (core/cutoff_embedder.py:151-214; cut_to_dist,
 * cutoff_shift, cutoff_inputs) and the cat of NeRF.encode_pts (core/networks/nerf.py:222-250), for the
 * samples [row0, row0 + nrows) of the R x S grid (either z [R,S] + rays, or pts [R*S,3]):
 *   x0   [nrows, (1+2L)*24 + 72] = [ (c-v)w | sin(2^l sh)w | cos(2^l sh)w ... (block-major, joint-minor) | unit dirs ]
 *   w_out[nrows, 24]             = 1 - sigmoid(tau (v - c))     (re-used by danbo_anerf_color_fwd) */
int danbo_anerf_encode_fwd(const float* rays_o, const float* rays_d, const float* z, const float* pts, int R, int S, int G,
                     const float* skts /*[G,24,4,4]*/, const float* align /*[24,4,4]*/, const float* cutoff /*[24]*/,
                     float tau, int L, long row0, int nrows, float* x0, float* w_out, void* stream);

/* transform_batch_rays (core/encoders.py:305-317) + VecNormEncoder + the sin/cos part of the view
 * CutoffEmbedder (dist_inputs; core/cutoff_embedder.py:156-166): E [R, (1+2L)*72], block-major. */
int danbo_anerf_view_pe_fwd(const float* rays_d, const float* skts, int R, int G, int L, float* E, void* stream);

/* view branch of NeRF.inference (core/networks/nerf.py:196-209) for the rays [ray0, ray0+nrays) whose
 * S samples are rows [0, nrays*S) of featv / w / alpha:
 *   x = relu(featv[row] + table[code(ray)] + sum_j w[row,j] C[j,ray,:]);  raw[ray,s] = (rgb_w x + rgb_b, alpha[row])
 * featv = (views_linears.0[:, :W] feature_linear) h  [rows,VW];  C [24,R_total,VW] = per-ray, per-joint
 * products of views_linears.0 with E;  table [n_codes+1,VW] = frame-code part + folded biases, last row =
 * mean code (Optcodes eval, core/networks/embedding.py:22-23);  cam_idx int64 [R_total] or NULL. */
int danbo_anerf_color_fwd(const float* featv, int ld_featv, const float* w, const float* C, const float* table,
                    const int64_t* cam_idx, int n_codes, int R_total, int ray0, int nrays, int S, int VW, const float* rgb_w,
                    const float* rgb_b, const float* alpha, int ld_alpha, float* raw_out /*[R_total,S,4]*/, void* stream);

/* ---------------------------------------------------------------------------------------------
 * One dense layer on rows that live in HBM:  y = act([x1 | x2] W^T + bias)  -- nn.Linear (+ ReLU) as the reference's
 * trunks apply it (core/networks/nerf.py:176-195: `h = relu(l(h))`, skip layers `cat([input, h])`), with fp32-accurate
 * products on the fp16 matrix cores (hi/lo split, see csrc/k_linear16.hip).  Used by the A-NeRF trunk (W = 448).
 * ------------------------------------------------------------------------------------------- */

/* dev tool (tools/micro_linear16.py --trace): int64[256] device buffer for s_memtime stamps of one wavefront, NULL = off */
int danbo_linear16_set_trace(void* buf);
/* size of the packed weight buffer of an N x (K1 + K2) layer; -1 if unsupported (N > 512) */
int danbo_linear16_packed_bytes(int N, int K1, int K2);
/* W[n, k] = w[n * stride_n + k * stride_k]  (strides in floats: (K, 1) for nn.Linear.weight, (1, N) for its transpose);
 * columns [0, K1) multiply x1, [K1, K1 + K2) multiply x2 */
int danbo_linear16_pack(const float* w, long stride_n, long stride_k, int N, int K1, int K2, void* packed, void* stream);
/* y [M, ldy] (first N columns written); x1 [M, ld1], x2 [M, ld2] or NULL (K2 = 0); bias [N] or NULL; act 0 = none,
 * 1 = ReLU; rows = min(*count, M) if count != NULL (device-side row count of a compacted list) */
int danbo_linear16_fwd(const float* x1, int ld1, int K1, const float* x2, int ld2, int K2, const void* packed,
                       const float* bias, int N, int act, float* y, int ldy, int M, const int32_t* count, void* stream);
/* The same layer with activations in FRAGMENT ORDER between the layers of a trunk (csrc/k_linear16.hip, Lin16Args::frag): a
 * [rows, C] activation, C % 32 == 0, lives in a buffer of round_up(rows, 128) * C floats laid out
 * [rows / 16][C / 32][2][4 (q)][16 (n)][4 (i)] <-> element (16 g + n, 32 s + 16 h + 4 q + i), so that every load / store
 * instruction of a wavefront moves one contiguous KB.  frag bits: 1 = x1, 2 = x2, 4 = y are fragment-order buffers (their ld is
 * ignored); a layer whose input part is in fragment order is packed with the matching bit of frag_in (bit 0: K1 part, bit 1: K2). */
int danbo_linear16_pack_frag(const float* w, long stride_n, long stride_k, int N, int K1, int K2, int frag_in, void* packed,
                             void* stream);
int danbo_linear16_fwd_frag(const float* x1, int ld1, int K1, const float* x2, int ld2, int K2, const void* packed,
                            const float* bias, int N, int act, float* y, int ldy, int M, const int32_t* count, int frag,
                            void* stream);

/* The first / the skip layer of the A-NeRF trunk with the 24 (1 + 2 L) + 72 density inputs RECOMPUTED in the kernel
 * (CutoffEmbedder._embed, core/cutoff_embedder.py:151-214) from the encoder's compact table [M, 144] that
 * danbo_anerf_encode_compact writes -- per joint (cutoff - distance, shifted distance, cutoff weight, direction x), then the 24
 * (direction y, z) pairs -- instead of read: 576 instead of 1 728 B per sample, written once and read twice; the layer keeps
 * its 14 k-steps.
 * danbo_linear16_pack_enc: w [N, 24 (1 + 2 L) + 72 + K2] (frag_in 2: the K2 part arrives in fragment order), packed:
 * danbo_linear16_packed_bytes(N, DANBO_LINEAR16_ENC_K, K2) bytes.  danbo_linear16_fwd_enc: y in fragment order, N = 448. */
#define DANBO_LINEAR16_ENC_K 448
#define DANBO_ANERF_ENC_FLOATS 144
int danbo_linear16_pack_enc(const float* w, long stride_n, long stride_k, int N, int L, int K2, int frag_in, void* packed,
                            void* stream);
int danbo_linear16_fwd_enc(const float* table, int L, const float* x2, int K2, const void* packed, const float* bias, int N,
                           int act, float* y, int M, const int32_t* count, void* stream);
int danbo_anerf_encode_compact(const float* rays_o, const float* rays_d, const float* z, const float* pts, int R, int S, int G,
                               const float* skts, const float* align, const float* cutoff, float tau, long row0, int nrows,
                               float* table, float* w_out, void* stream);
/* A-NeRF's head layer with the COLOUR HEAD AS ITS EPILOGUE (round 6; replaces danbo_linear16_fwd_frag + danbo_anerf_color_fwd: the
 * (VW + 1)-wide head rows are never written).  x1: the trunk's last activation in fragment order [M, K1]; packed / bias: the layer
 * [views_linears.0[:, :W] feature_linear ; alpha_linear] packed with frag_in 1 (N = VW + 1: VW view features, then the density
 * logit); its M rows are the samples of rays [ray0, ray0 + M / S), ray-major -- the 16 rows of a wavefront are 16 samples of ONE
 * ray, so the cutoff-weighted sum over the joints is one more k-step of the same GEMM (A = the ray's joint vectors C[:, ray, :],
 * B = the rows' cutoff weights w [M, 24]; the row of `table` the ray's camera selects rides along as a 25th joint of weight 1),
 * followed by ReLU, rgb_linear and the store of raw_out [R_total, S, 4] (reference core/networks/nerf.py:196-209).
 * w, C [24, R_total, VW], table [n_codes + 1, VW], cam_idx, rgb_w, rgb_b: as danbo_anerf_color_fwd.  S % 16 == 0, VW % 16 == 0,
 * VW <= 240, K1 % 32 == 0. */
int danbo_linear16_fwd_color(const float* x1, int K1, const void* packed, const float* bias, int VW, int M, const float* w,
                             const float* C, const float* table, const int64_t* cam_idx, int n_codes, int R_total, int ray0, int S,
                             const float* rgb_w, const float* rgb_b, float* raw_out, void* stream);


/* ---------------------------------------------------------------------------------------------
 * Training step (SURVEY 8b "+ _bwd"; reference core/trainer.py:257-302,563-576 = forward, loss, loss.backward(), Adam).
 * The density / colour MLP runs on the compacted in-volume rows as
 *   forward   danbo_trunk_fwd   one register-resident chain per pass (activations written once, in fragment order)
 *   backward  danbo_trunk_bwd   the mirrored input-gradient chain dz_7 .. dz_0 -> d h
 *             danbo_dw16        dW_l = dz_l^T x_l, db_l = sum_rows dz_l: all layers in one launch, split over row slices
 * (declared further down), all with fp32-accurate products on the fp16 matrix cores (hi/lo split).  Gradients are ~1e-6:
 * every backward GEMM pre-scales its gradient operand by a power of two (per 16-row group in the chain; from the running
 * max |.| the chain recorded in the weight-gradient kernel) and scales the result back -- exact, and it keeps the lo halves
 * out of fp16's subnormals.
 * ------------------------------------------------------------------------------------------- */
/* dW / db of up to 13 dense layers in one launch pair (csrc/k_dw16.hip).  dy [rows, ldy]: gradient with respect to the layer's
 * pre-activation; x1 | x2: the layer's input(s); gw [N, K1 + K2] and gb [N] in nn.Linear layout are OVERWRITTEN.  A layer
 * evaluated for two parameters at once (feature_linear + alpha_linear) writes rows >= split_n to gw2 / gb2.  16-byte aligned
 * bases, row strides multiples of 4 floats, x1 readable up to a multiple of 4 columns.  rows = min(*count, M). */
typedef struct DanboDwLayer {
    const float *dy, *x1, *x2, *dy_maxabs /* device scalar max |dy| recorded by the producer, or NULL */;
    float *gw, *gw2, *gb, *gb2;
    int ldy, ld1, ld2, N, K1, K2, split_n;
    int frag;                 /* bit 0: dy, bit 1: x1 (then K2 = 0) are fragment-order buffers (widths multiples of 32, ld ignored) */
    int gw_ld, gw_col0;       /* 0, 0: gw is [N, K1 + K2]; else the gradient goes to gw[n * gw_ld + gw_col0 + k] -- a layer with two
                                 inputs in different layouts is passed as two layers (gb = NULL in one of them) */
    int x1_pe;                /* 1: x1 is the fused trunk's `pe` buffer (fragment order, K1 = 224 slots): slot k is column
                                 danbo_trunk_pe_column(k) of the gradient, padding slots are dropped */
} DanboDwLayer;
long danbo_dw16_scratch_floats(const DanboDwLayer* layers, int n_layers, int slices);
int danbo_dw16(const DanboDwLayer* layers, int n_layers, int M, const int32_t* count, int slices, float* scratch, void* stream);

/* ---- building blocks of the training step (csrc/k_train_rows.hip, k_assign_bwd.hip, k_pose_bwd.hip); danbo_train_step
 * below enqueues all of them.  Row layout and the device counters cnt[] are described at the top of k_train_rows.hip. ---- */

/* K1b + K2 forward for the training step: as danbo_gather_assign_blend16_fwd on rows [*first, *count) of list / h, and
 * h[row][15] = q = sum_j p_j valid_j instead of the zero pad (the row's assignment mass of the soft-softmax loss) */
int danbo_gather_assign_blend16_train(const float* rays_o, const float* rays_d, const float* z, int R, int S, int G,
                                      const float* skts, const float* align, const float* axis_scale, const float* volumes,
                                      const uint32_t* valid_bits, const int32_t* list, const int32_t* count, const int32_t* first,
                                      int n, const void* packed16, const float* b0, const float* b1, const float* w2,
                                      const float* b2, float* h, uint32_t* ticket, void* stream);
/* per-ray view inputs [PE(dir) | frame code | 0] (nerf.py:252-279); ray_mode / normalise as danbo_view_consts */
int danbo_train_view_inputs(const float* rays_d, const float* skts, int R, int G, int ray_mode, int normalise, int L_view,
                            const float* codes, int n_codes, int Cf, const int64_t* cam_idx, float* vin, int ldv, void* stream);
/* d loss / d (rgb_map, acc_map) of both passes for L1 (mse = 0) or MSE on rgb + (1 - acc) bg (trainer.py:396-422);
 * loss[0] += fine term, loss[1] += coarse term */
int danbo_train_loss_grad(const float* rgb, const float* acc, const float* rgb0, const float* acc0, const float* target,
                          const float* bgs /*[R,3] or NULL (1)*/, int use_bg, int R, int mse, float w_fine, float w_coarse,
                          float* g_rgb, float* g_acc, float* g_rgb0, float* g_acc0, float* loss, void* stream);
/* un-merge d raw of the final composite onto the two passes' samples (+ the coarse composite's own gradient), ray sums of the
 * samples outside every volume -> d_raw_rows[ray], soft-softmax labels (trainer.py:507-536), loss[2] += labels outside */
int danbo_train_draw_unmerge(float* d_raw_c, const float* d_raw_sorted, const int32_t* order, const uint32_t* bits_c,
                             const uint32_t* bits_f, const float* weights, const float* alpha, int R, int S, int Sf,
                             float* d_raw_f, float* d_raw_rows, uint8_t* label_c, uint8_t* label_f, float* loss, float* maxabs,
                             void* stream);
/* ABI 7: the three calls above -- loss gradients, the adjoints of the coarse and of the merged composite (danbo_composite_bwd_lazy
 * twice) and the un-merge -- as ONE launch, one wavefront per ray, the composites' d raw held in LDS in between; every output is the
 * separate calls' bit for bit (the loss sums are atomics either way).  S + Sf <= 256.  This is what danbo_train_step runs. */
int danbo_train_mid(const float* rgb, const float* acc, const float* rgb0, const float* acc0, const float* target, const float* bgs,
                    int use_bg, int R, int S, int Sf, int mse, float w_fine, float w_coarse, float density_scale, float* g_rgb,
                    float* g_acc, float* g_rgb0, float* g_acc0, const float* raw_c, const float* raw_empty, const float* raw_sorted,
                    const uint32_t* bits_c, const uint32_t* bits_f, const float* z_c, const float* z_sorted, const float* rays_d,
                    const float* noise_c /*or NULL*/, const float* noise_f /*or NULL*/, const int32_t* order, const float* weights,
                    const float* alpha, float* d_raw_c, float* d_raw_f, float* d_raw_rows, uint8_t* label_c, uint8_t* label_f,
                    float* loss, float* maxabs, void* stream);
int danbo_train_bone_lists(const uint32_t* bits_c, const uint32_t* bits_f, const int32_t* row_sample, const int32_t* cnt, int R,
                           int rows_cap, int32_t* lists /*[24, rows_cap]*/, int32_t* cntb /*[24], zeroed by the caller*/, void* stream);

/* backward of K1b + K2 with the forward recomputed from the pose volumes (csrc/k_assign_bwd.hip) */
typedef struct DanboAssignBwd {
    const float *rays_o, *rays_d, *z_c /*[R,S]*/, *z_f /*[R,Sf]*/, *skts, *align, *axis_scale, *volumes;
    int R, S, Sf, G, rows_cap;
    const int32_t *row_sample, *row_ray, *cnt, *lists, *cntb;
    const float *h_rows /*[rows,16], [15] = q*/, *d_h /*[rows,16]*/;
    const uint8_t *label_c, *label_f;
    const uint32_t *bits_c, *bits_f;
    const float *w0, *adj_w, *adj, *b0, *w1, *b1, *w2, *b2;          /* prob_linears.* in the reference's layouts */
    float *g_w0, *g_adj_w, *g_b0, *g_w1, *g_b1, *g_w2, *g_b2;       /* accumulated: the caller zeroes them */
    float *g_vol /*[G,24,240]*/, *g_scale /*[24,3]*/;
    float c_ss;        /* 2 soft_softmax_loss_coef / (R (S + Sf)) */
    float* loss;       /* loss[2] += (label - q)^2 of the in-volume rows */
    const float* d_p;  /* ABI 3: [rows,24] upstream gradient of the masked probabilities p_j valid_j (added to d h . f_j), or NULL */
} DanboAssignBwd;
int danbo_assign_blend_bwd(const DanboAssignBwd* p, void* stream);

/* backward of danbo_pose_volumes_fwd (csrc/k_pose_bwd.hip).  adj_w / adj: the raw parameter and the 0/1 buffer [24,24];
 * fwd_scratch: what the forward left in its scratch; g_w*, g_b2, g_b3 are overwritten, g_adj_w*, g_b0, g_b1 accumulated. */
int danbo_pose_volumes_bwd(const float* bones, int G, int L_graph, int W, const float* w0, const float* adj_w0, const float* adj0,
                           const float* b0, const float* w1, const float* adj_w1, const float* adj1, const float* b1,
                           const float* w2, const float* w3, const float* fwd_scratch, const float* d_volumes, float* g_w0,
                           float* g_adj_w0, float* g_b0, float* g_w1, float* g_adj_w1, float* g_b1, float* g_w2, float* g_b2,
                           float* g_w3, float* g_b3, float* bwd_scratch /*>= 2 G 24 W floats*/, void* stream);

/* The random draws of a training step in one launch (csrc/k_step_io.hip) -- what the reference draws with torch.rand
 * (sample_from_lineseg / isample_from_lineseg, core/utils/ray_utils.py:206-291) and torch.randn * raw_noise_std (NeRF.raw2outputs,
 * core/networks/nerf.py:316): n_uniform floats in [0, 1) (24 bits) and n_normal floats ~ N(0, normal_std^2) from Philox4x32-10.
 * state (device, 3 x uint64): [0] seed, [1] counter of the next call, [2] 0 between calls.  The KERNEL advances the counter, so a
 * captured HIP graph draws fresh numbers on every replay; word j of either stream is a pure function of (seed, counter, j):
 * uniform[4 i + e] = (philox(counter + i, 0 | seed)[e] >> 8) * 2^-24 (tests/test_gpu_train_engine.py restates it). */
int danbo_random_draws(uint64_t* state, long n_uniform, float* uniform, long n_normal, float normal_std, float* normal,
                       void* stream);

/* Rows of up to DANBO_MAX_ROW_SPANS tensors gathered into one flat buffer in one launch (the captured training step's static
 * inputs; the reference's loader hands over per-ray copies of the per-pose tensors, core/trainer.py:257-302, of which every
 * (R / G)-th row is taken): span i copies rows x row_words 32-bit words, row r from src + r * src_row_stride_words, to
 * dst + dst_word (contiguous).  `spans` is a HOST array, read during the call. */
#define DANBO_MAX_ROW_SPANS 12
typedef struct DanboRowSpan {
    const void* src;
    long dst_word, src_row_stride_words;
    int rows, row_words;
} DanboRowSpan;
int danbo_gather_rows(const DanboRowSpan* spans, int n_spans, void* dst, void* stream);

/* torch.optim.Adam's update (amsgrad = False, weight_decay = 0; core/raycasters.py:75) on a flat parameter buffer.
 * The step's scalars travel BY VALUE as kernel arguments (ABI 2; ABI 1 read them from a device buffer the host had to keep
 * alive and unchanged until the kernel ran): lr, bias_corr1 = 1 - beta1^t, sqrt_bias_corr2 = sqrt(1 - beta2^t),
 * grad_scale = 1 / world size. */
int danbo_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long n, float lr, float bias_corr1,
                    float sqrt_bias_corr2, float grad_scale, float beta1, float beta2, float eps, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Fused trunk of the training step: the density / colour MLP of core/networks/nerf.py:176-209 over the compacted rows of a
 * step as ONE forward kernel per pass (csrc/k_mlp16.hip: K3's register-resident chain, which also writes every activation the
 * backward needs, once, in the lanes' own order) and ONE input-gradient kernel (csrc/k_mlp16_bwd.hip: the mirrored chain
 * dz_7 .. dz_0 with the adjoint of the positional encoding at its end).  feature_linear and the per-sample part of
 * views_linears.0 are evaluated merged (W_fv = W_v[:, :256] W_f, as in danbo_mlp16_pack); the weight-gradient kernel
 * produces dW_fv, from which danbo_train_head_chain derives the gradients of both layers (parameter-sized GEMMs).
 * Row layout and counters: csrc/k_train_rows.hip.  Fragment order of a [rows, C] buffer: [rows/16][C/32][2][64 lanes][4 floats].
 * ------------------------------------------------------------------------------------------- */
#define DANBO_TRUNK_FWD_CHUNKS 74
#define DANBO_TRUNK_BWD_CHUNKS 76
#define DANBO_TRUNK_PACKED_BYTES ((DANBO_TRUNK_FWD_CHUNKS + DANBO_TRUNK_BWD_CHUNKS) * 32768)
#define DANBO_TRUNK_PE_WIDTH 224    /* 7 k-steps x 32 slots; slot -> pts_linears column: danbo_trunk_pe_column */

typedef struct DanboTrunkWeights {
    const float* pts_w[8];             /* pts_linears.{0..7}.weight [256,195 | 256 | 451] */
    const float* pts_b[8];
    const float *alpha_w, *alpha_b, *feature_w, *feature_b, *views_w /*[128,256+view_ch]*/, *views_b, *rgb_w, *rgb_b;
    int view_ch;
    /* written by danbo_trunk_pack */
    void* packed;                      /* DANBO_TRUNK_PACKED_BYTES */
    float* wfv;                        /* [128,256] W_fv */
    float* b_eff;                      /* [128] views_b + W_v[:, :256] feature_b */
    float* wmax;                       /* [16]: max |w| of pts_linears.0..7, W_fv -- ZEROED by the caller before the call */
    float* winv;                       /* [16]: exact inverses of the power-of-two pack scales, same order */
} DanboTrunkWeights;
int danbo_trunk_pack(const DanboTrunkWeights* w, void* stream);

typedef struct DanboTrunkRows {
    int32_t* cnt;                      /* the step's device counters (danbo_trunk_fwd derives [1]..[7] from [0]) */
    const int32_t* row_sample;         /* [rows_cap] sample index of row i >= R inside its pass */
    const float* h_rows;               /* [rows_cap,16] blended features (rows < R are not read: empty-space rows have h = 0) */
    const float* cview;                /* [R,128] per-ray part of the view layer incl. b_eff (danbo_train_cview) */
    int R, S, Sf, rows_cap;
    long rows_pad;                     /* rows of every fragment-order buffer: rows_cap rounded up to 128, + 128 */
    /* forward -> backward, weight gradients */
    float* y;                          /* [8][rows_pad*256] post-ReLU activations of pts_linears.0..7 */
    float* pe;                         /* [rows_pad*224] positional encoding in the lanes' slot order */
    uint64_t* relu;                    /* [8][rows_pad*4] sign bits: bit 4 T + i of (row, q) <-> column 16 T + 4 q + i */
    float* hv;                         /* [rows_pad*128] post-ReLU view layer */
    uint32_t* hv_bits;                 /* [rows_pad*4] */
    float* raw_rows;                   /* [rows_cap,4] */
    float *raw_c /*[R*S,4]*/, *raw_f /*[R*Sf,4]*/, *raw_empty /*[R,4]*/;
    int32_t* row_ray;                  /* [rows_cap] */
    /* backward */
    const float *d_raw_c, *d_raw_f;    /* dense gradients of the two passes' raw; both NULL: d_raw_rows holds d raw of EVERY row */
    float* d_raw_rows;                 /* [rows_cap,4]: rows < R hold the rays' empty-space sums on entry; rows >= R are written */
    float* dz;                         /* [8][rows_pad*256] gradients with respect to the pre-activations */
    float* dpre_v;                     /* [rows_pad*128] */
    float* d_alpha4;                   /* [rows_cap,4]: (d alpha, 0, 0, 0) */
    float* d_h;                        /* [rows_cap,16] */
    float* maxabs;                     /* [16]: running max |dz_0..7|, |dpre_v|, |d alpha| -- ZEROED by the caller */
} DanboTrunkRows;
int danbo_trunk_fwd(const DanboTrunkWeights* w, const DanboTrunkRows* r, int pass, void* stream);
int danbo_trunk_bwd(const DanboTrunkWeights* w, const DanboTrunkRows* r, void* stream);
/* pts_linears.0 column (0..194) of slot k (0..223) of the `pe` buffer, or -1 for the padding slots (host helper) */
int danbo_trunk_pe_column(int k);

/* the view branch around the fused trunk (csrc/k_train_head.hip): cview [R,128] = W_v[:, 256:] vin + b_eff */
int danbo_train_cview(const float* vin /*[R,ldv]*/, int ldv, int view_ch, const float* views_w, const float* b_eff, int R, float* cview,
                      void* stream);
/* d cview[ray] = sum over the ray's rows of d pre_v (fragment order); g_views_w[:, 256:] += d cview^T vin; csum[cam] = sum over
 * the camera's rays of d cview.  d_cview and csum must be zero on entry. */
int danbo_train_view_grads(const float* dpre_v, const int32_t* row_ray, const int32_t* cnt, int rows_cap, int R, const float* vin, int ldv,
                           int view_ch, const int64_t* cam_idx, int n_codes, float* d_cview, float* csum, float* g_views_w,
                           float* vg_part, void* stream);
/* vg_part (ABI 3): DANBO_TRAIN_VG_PART_FLOATS floats or NULL.  With it the per-ray-slice partial sums of d W_v[:, 256:] are
 * stored there and danbo_train_head_chain adds them to g_views_w in a fixed order; without it they are accumulated with atomics. */
#define DANBO_TRAIN_VG_PART_FLOATS (32 * 128 * 160)
/* gradients of feature_linear, views_linears.0[:, :256] / bias and the frame codes from d W_fv, d b_eff (k_dw16) and csum */
int danbo_train_head_chain(const float* g_wfv, const float* g_beff, const float* csum, const float* feature_w, const float* feature_b,
                           const float* views_w, int view_ch, int n_codes, int code_size, int code_col0, float* g_feature_w,
                           float* g_feature_b, float* g_views_w, float* g_views_b, float* g_codes, const float* vg_part, void* stream);

/* ---------------------------------------------------------------------------------------------
 * One training batch behind one call: forward, losses, backward (csrc/k_train.hip) -- what Trainer.train_batch
 * (core/trainer.py:257-302) does between dict_to_device and optimizer.step(), for the shipped DANBO structure
 * (FGNNcat + vox_MIXGNN + sigmoid, D = 8, W = 256, skip after layer 4, view_W = 128, single_net).
 * Parameters and gradients are handed over as pointers into two flat fp32 buffers (g_flat is zeroed by the call).
 * ------------------------------------------------------------------------------------------- */
enum DanboTrainTensor {
    DANBO_T_G_W0 = 0, DANBO_T_G_ADJW0, DANBO_T_G_B0, DANBO_T_G_W1, DANBO_T_G_ADJW1, DANBO_T_G_B1, DANBO_T_G_W2, DANBO_T_G_B2,
    DANBO_T_G_W3, DANBO_T_G_B3, DANBO_T_AXIS_SCALE,                       /* graph_net.layers.{0..3}.*, graph_net.axis_scale */
    DANBO_T_A_W0, DANBO_T_A_ADJW, DANBO_T_A_B0, DANBO_T_A_W1, DANBO_T_A_B1, DANBO_T_A_W2, DANBO_T_A_B2,   /* prob_linears.layers.* */
    DANBO_T_PTS_W0, DANBO_T_PTS_B0 = DANBO_T_PTS_W0 + 8,                  /* pts_linears.{0..7}.{weight, bias} */
    DANBO_T_ALPHA_W = DANBO_T_PTS_B0 + 8, DANBO_T_ALPHA_B, DANBO_T_FEAT_W, DANBO_T_FEAT_B, DANBO_T_VIEWS_W, DANBO_T_VIEWS_B,
    DANBO_T_RGB_W, DANBO_T_RGB_B, DANBO_T_CODES,                          /* framecodes.codes.weight (unused when n_codes = 0) */
    DANBO_T_COUNT
};

typedef struct DanboTrainModel {
    const float* p[DANBO_T_COUNT];     /* parameters, the reference's layouts */
    float* g[DANBO_T_COUNT];           /* their gradients (inside g_flat); alpha_linear.bias must follow feature_linear.bias */
    float* g_flat;                     /* the flat gradient buffer, zeroed at the start of the step */
    long n_flat;
    const float *g_adj0, *g_adj1, *a_adj;   /* the 0/1 adjacency buffers [24,24] of the three graph-conv layers */
    const float *align /*[24,4,4]*/, *init_scale /*[24,3]: graph_net.init_scale*/;
    int L_graph, graph_width, L_view, L_voxel, ray_mode, normalise, n_codes, code_size;
    int view_ch;                       /* 3 (1 + 2 L_view) + code_size: per-sample view inputs of views_linears.0 */
    int use_volume_near_far, loss_mse /*0: L1*/, use_background;
    float density_scale, rgb_loss_coef, coarse_weight, soft_softmax_coef, vol_scale_penalty /*0 unless opt_vol_scale*/;
} DanboTrainModel;

typedef struct DanboTrainBatch {
    const float *rays_o, *rays_d /*[R,3]*/, *skts /*[G,24,4,4]*/, *bones /*[G,24,3]*/, *cyls /*[G,5]*/;
    const float *near_in, *far_in /*[R] placeholders or NULL (0 / 1)*/;
    const int64_t* cam_idx /*[R]*/;
    const float *target /*[R,3]*/, *bgs /*[R,3] or NULL (white)*/;
    /* the step's random draws (NULL = the deterministic eval choice): stratified offsets [R,S], inverse-CDF uniforms [R,Sf],
     * density noise of the two composites [R,S] / [R,S+Sf], already multiplied by raw_noise_std * B (nerf.py:316) */
    const float *t_rand, *u_rand, *noise_c, *noise_f;
    int R, G, S, Sf, chunk;
    /* ABI 7: rng_state != NULL makes the step draw its own numbers -- danbo_random_draws(rng_state, n_uniform, rng_uniform, n_normal,
     * normal_std, rng_normal) on the caller's stream BEHIND the fork of its prologue branches (launched by the caller in front of the
     * step, the 13 us of the generator delayed every branch); t_rand / u_rand / noise_c / noise_f then point into rng_uniform /
     * rng_normal.  Phase 2 of a split step draws nothing. */
    uint64_t* rng_state;
    float *rng_uniform, *rng_normal;
    long long n_uniform, n_normal;
    float normal_std;
} DanboTrainBatch;

typedef struct DanboTrainOut {
    float *rgb_map /*[R,3]*/, *disp_map, *acc_map /*[R]*/, *alpha, *weights /*[R,S+Sf]*/, *rgb0, *disp0, *acc0, *alpha0 /*[R,S]*/;
    float* loss;      /* [4]: rgb loss, coarse rgb loss, sum over ALL samples of (label - q)^2 (x coef / (R (S+Sf)) = soft-softmax
                         loss), volume-scale loss */
    int32_t* counts;  /* [8] or NULL: row counters of the step ([1] coarse in-volume samples, [3] importance ones) */
} DanboTrainOut;

size_t danbo_train_workspace(const DanboTrainModel* model, int R, int G, int S, int Sf, int chunk);
/* S >= 3, S + Sf <= 256.  Enqueues ~75 kernels on `stream`; nothing synchronises, every data-dependent size stays on the device. */
int danbo_train_step(const DanboTrainModel* model, const DanboTrainBatch* batch, const DanboTrainOut* out, void* workspace,
                     size_t workspace_bytes, void* stream);
/* the same step in two pieces for data-parallel training: phase 1 = everything up to and including the pose-GNN adjoint (after
 * it every gradient except the dense layers' -- pts_linears.*, alpha / feature / views / rgb_linear -- is final and can go into
 * the all-reduce), phase 2 = the dense layers' weight gradients (danbo_dw16) + the loss copy; phase 0 = both.  Same arguments
 * for both calls. */
int danbo_train_step_phase(const DanboTrainModel* model, const DanboTrainBatch* batch, const DanboTrainOut* out, void* workspace,
                           size_t workspace_bytes, int phase, void* stream);

/* Where a finished step left its sampling decisions inside `workspace` (device pointers, valid until the next step on it): the
 * depths and merge order RayCaster.render_rays forms and does not return (core/raycasters.py:310-311,346-361: z_vals,
 * z_samples, sorted_idxs) and the in-volume words of both passes.  Host-only (no launch); for callers that want to re-evaluate
 * the same samples elsewhere (the parity tests hand them to a float64 reference).  0 on success. */
typedef struct DanboTrainView {
    const float *z_coarse /*[R,S]*/, *z_fine /*[R,Sf]*/, *z_sorted /*[R,S+Sf]*/;
    const int32_t* order /*[R,S+Sf]: index into [coarse | fine]*/;
    const uint32_t *bits_coarse /*[R,S]*/, *bits_fine /*[R,Sf]*/;
} DanboTrainView;
int danbo_train_workspace_view(const DanboTrainModel* model, int R, int G, int S, int Sf, int chunk, void* workspace, DanboTrainView* view);

/* ---------------------------------------------------------------------------------------------
 * A-NeRF (nerf_type = nerf) on this library's kernels end to end (ABI 8; csrc/k_anerf_train.hip).
 * Render: the per-ray, per-joint view constants.  Training: Trainer.train_batch (core/trainer.py:257-302) for NeRF with the
 * cutoff encoders (core/networks/nerf.py:107-122,176-209,222-279; core/cutoff_embedder.py:151-214) -- every dense W-wide layer on
 * danbo_linear16_fwd both ways and danbo_dw16, everything else on the kernels declared here; no library GEMM.
 * ------------------------------------------------------------------------------------------- */

/* wj [24][3 (1 + 2 L)][VW]: the view columns of views_linears.0.weight ([VW, ld]; columns col0 + 72 b + 3 j + a, b = encoding
 * block, a = axis) regrouped per joint, wj[j][3 b + a][c] */
int danbo_anerf_view_wj_pack(const float* views_w, int ld, int col0, int VW, int L, float* wj, void* stream);
/* C [24, R, VW]: C[j, ray, :] = sum_kk wj[j][kk][:] E[ray, j, kk] with E = the cutoff view encoding before its per-sample weight
 * (danbo_anerf_view_pe_fwd's values: transform_batch_rays + VecNormEncoder + sin / cos, core/encoders.py:305-317,774-795,
 * core/cutoff_embedder.py:156-166), formed in the kernel -- what core/anerf_engine.py computed with torch.bmm until round 5.
 * One fmaf chain per output, kk ascending. */
int danbo_anerf_view_consts_fwd(const float* rays_d, const float* skts, int R, int G, int L, const float* wj, int VW, float* C,
                                void* stream);
/* adjoint: g_views_w[c * ld + col0 + 72 b + 3 j + a] = sum_rays E[ray, j, 3 b + a] dC[j, ray, c] (OVERWRITTEN; ray slices summed in
 * a fixed order); scratch: danbo_anerf_view_consts_bwd_scratch_floats(R, L, VW) floats */
long danbo_anerf_view_consts_bwd_scratch_floats(int R, int L, int VW);
int danbo_anerf_view_consts_bwd(const float* rays_d, const float* skts, int R, int G, int L, const float* dC, int VW, float* g_views_w,
                                int ld, int col0, float* scratch, void* stream);
/* danbo_anerf_color_fwd for the rays [0, nrays) of a training pass (rows ray * S + s): table_ray [R_total, VW] holds each ray's own
 * bias + frame-code row (danbo_anerf_ray_table), hv [rows, VW] = relu(pre_v) is kept for the backward */
int danbo_anerf_color_train_fwd(const float* featv, int ld_featv, const float* w, const float* C, const float* table_ray, int R_total,
                                int nrays, int S, int VW, const float* rgb_w, const float* rgb_b, const float* alpha, int ld_alpha,
                                float* hv, float* raw_out, void* stream);
/* its adjoint.  d_raw [rows, 4] -> d_featv [rows, VW] and (d alpha, 0, 0, 0) at d_alpha_out + row * ld_dalpha (16-byte blocks), both
 * MULTIPLIED by the power of two sigma that brings *raw_max (device: max |d raw|) to [64, 128) -- the operands of the fp16-split
 * backward GEMMs -- with sigma written to *sig_top; dC [24, R_total, VW] and d_pre_ray [R_total, VW] (sums over the ray's samples of
 * w_j g and of g, g = d pre_v) unscaled, overwritten or (accumulate) added to; part: danbo_anerf_color_bwd_part_floats(nrays, VW)
 * floats of per-wavefront sums for danbo_anerf_rgb_reduce (-> d rgb_linear.weight / .bias, passes concatenated) */
long danbo_anerf_color_bwd_part_floats(int nrays, int VW);
int danbo_anerf_color_bwd(const float* d_raw, const float* hv, const float* w, int R_total, int nrays, int S, int VW, const float* rgb_w,
                          const float* raw_max, float* sig_top, float* d_featv, float* d_alpha_out, int ld_dalpha, float* dC,
                          float* d_pre_ray, int accumulate, float* part, void* stream);
int danbo_anerf_rgb_reduce(const float* part, long part_floats, int VW, float* g_rgb_w, float* g_rgb_b, void* stream);
/* table_ray[ray] = views_linears.0.bias + views_linears.0.weight[:, code0 : code0 + code_size] codes[cam_idx[ray]] (Optcodes in
 * training mode, core/networks/embedding.py:24-39; code_size = 0: the bias) and the adjoint: the code columns' and the bias'
 * gradient, framecodes.codes.weight's rows (g_codes zero on entry; rays of one camera added in ray order, no atomics) */
int danbo_anerf_ray_table(const float* views_w, int ld, int code0, int code_size, const float* views_b, const float* codes, int n_codes,
                          const int64_t* cam_idx, int R, int VW, float* table_ray, void* stream);
int danbo_anerf_code_grads(const float* d_pre_ray, const float* views_w, int ld, int code0, int code_size, const float* codes,
                           int n_codes, const int64_t* cam_idx, int R, int VW, float* g_views_w, float* g_views_b, float* g_codes,
                           float* v_scratch /*[R, code_size]*/, void* stream);
/* dz = t . [y > 0] . rho, n floats (n % 4 == 0): rho = 1 (prev_max NULL) or the power of two that brings *prev_max to [64, 128);
 * *sig_out = *sig_in * rho; *max_out = max(*max_out, max |dz|) */
int danbo_anerf_relu_mask(const float* t, const float* y, long n, const float* prev_max, const float* sig_in, float* sig_out,
                          float* max_out, float* dz, void* stream);
/* d_all [R (S + Sf), 4]: rows r S + s (coarse) then R S + r Sf + s (importance) <- d_sorted [R, S + Sf, 4] through `order`
 * (+ d_c0 [R, S, 4], the coarse composite's own gradient); *max_out = max |d_all| */
int danbo_anerf_unmerge(const float* d_sorted, const float* d_c0, const int32_t* order, int R, int S, int Sf, float* d_all,
                        float* max_out, void* stream);
/* danbo_anerf_encode_fwd with tau read from DEVICE memory when the kernel runs (the module's `tau` buffer: update_tau changes it
 * every step, core/cutoff_embedder.py:221-223, and a by-value argument is frozen inside a captured graph) */
int danbo_anerf_encode_fwd_dtau(const float* rays_o, const float* rays_d, const float* z, const float* pts, int R, int S, int G,
                                const float* skts, const float* align, const float* cutoff, const float* tau_dev, int L, long row0,
                                int nrows, float* x0, float* w_out, void* stream);

/* C [M, N] (row stride ldc) = A B (+ bias[n]) with A[m, k] = A[m sa_m + k sa_k], B[k, n] = B[k sb_k + n sb_n]: float32 operands,
 * the sum over k in ascending order accumulated in float64, rounded once.  For the parameter-sized products of a weight refresh
 * (folded head matrices, per-camera tables); one thread per output. */
int danbo_small_matmul(const float* A, long sa_m, long sa_k, const float* B, long sb_k, long sb_n, const float* bias, int M, int N, int K,
                       float* C, long ldc, void* stream);

#define DANBO_ANERF_MAX_D 8
typedef struct DanboAnerfTrainModel {
    int D, W, VW, skip /* pts_linears[skip + 1] takes [density inputs | h]; -1: none */, L, L_view, n_codes, code_size /*0: no frame codes*/;
    const float *pts_w[DANBO_ANERF_MAX_D], *pts_b[DANBO_ANERF_MAX_D];
    const float *alpha_w, *alpha_b, *feature_w, *feature_b, *views_w /*[VW, W + 72 (1 + 2 L_view) + code_size]*/, *views_b, *rgb_w, *rgb_b,
        *codes;
    float *g_pts_w[DANBO_ANERF_MAX_D], *g_pts_b[DANBO_ANERF_MAX_D];
    float *g_alpha_w, *g_alpha_b, *g_feature_w, *g_feature_b, *g_views_w, *g_views_b, *g_rgb_w, *g_rgb_b, *g_codes;
    float* g_flat;                 /* the flat gradient buffer all g_* point into: zeroed at the start of the step */
    long n_flat;
    const float *align /*[24,4,4]*/, *cutoff /*[24]: pe_fn.cutoff_dist*/, *tau /*DEVICE scalar: pe_fn.tau*/;
    int loss_mse /*0: L1*/, use_background;
    float density_scale, rgb_loss_coef, coarse_weight;
} DanboAnerfTrainModel;
/* One A-NeRF training batch: forward (bounds, stratified depths, encoders, trunk, heads, composite, importance depths, second pass,
 * merged composite), the two rgb losses, backward into g_flat.  DanboTrainBatch / DanboTrainOut as danbo_train_step (bones unused;
 * loss[2] = loss[3] = 0; counts ignored).  Enqueues ~110 kernels on `stream`, nothing synchronises: the step can be captured into
 * a HIP graph.  S >= 3, S + Sf <= 256, W % 4 == 0, W <= 508, VW % 4 == 0, VW <= 256. */
size_t danbo_anerf_train_workspace(const DanboAnerfTrainModel* model, int R, int G, int S, int Sf, int chunk);
int danbo_anerf_train_step(const DanboAnerfTrainModel* model, const DanboTrainBatch* batch, const DanboTrainOut* out, void* workspace,
                           size_t workspace_bytes, void* stream);
/* the step's sampling decisions inside `workspace` (as danbo_train_workspace_view; bits_* are NULL: A-NeRF has no in-volume mask) */
int danbo_anerf_train_workspace_view(const DanboAnerfTrainModel* model, int R, int G, int S, int Sf, int chunk, void* workspace,
                                     DanboTrainView* view);

/* ---------------------------------------------------------------------------------------------
 * The whole eval chain of one ray batch behind one call: RayCaster.render_rays (core/raycasters.py:245-377) with the DANBO
 * network (core/networks/danbo.py) -- what a C host binds instead of the reference's caster(ray_batch, ...) call.
 * Every pointer is a device pointer; the weight buffers are the ones the *_pack entry points above produce.
 * ------------------------------------------------------------------------------------------- */
typedef struct DanboModel {
    /* FactorizeGNN (danbo_pose_volumes_fwd) */
    const float *g_w0, *g_adjw0, *g_b0, *g_w1, *g_adjw1, *g_b1, *g_w2, *g_b2, *g_w3, *g_b3;
    int L_graph, graph_width;
    /* geometry */
    const float *align /*[24,4,4]*/, *axis_scale /*[24,3]*/;
    /* assignment net (danbo_assign16_pack) */
    const void* assign16;
    const float *a_b0, *a_b1, *a_w2, *a_b2;
    /* density / colour MLP (ABI 9: packed by danbo_mlp32_pack -- danbo_render_frame runs danbo_pe_mlp32_fwd) */
    const void* mlp16;
    const float* pts_b[8];
    const float *alpha_w, *alpha_b, *rgb_w, *rgb_b;
    /* view branch (danbo_mlp_pack -> views_w_ray_t, danbo_mlp16_pack -> views_b_eff, danbo_view_code_table) */
    const float *views_w_ray_t, *views_b_eff, *framecodes, *mean_code, *code_table, *empty_consts;
    int n_codes, code_size, L_view, ray_mode, normalise;
    float density_scale;
    int use_volume_near_far;
    int flat_rays_ok;   /* the caller's statements (a) and (b) of danbo_flat_rays about these weights: 1 = rays that cannot meet a
                           volume get their constants without view constants, resampling or composites; 0 = every ray is evaluated */
} DanboModel;

typedef struct DanboRays {
    const float *rays_o, *rays_d /*[R,3]*/, *skts /*[G,24,4,4]*/, *bones /*[G,24,3]*/, *cyls /*[G,5]*/;
    const int64_t* cam_idx /*[R] or NULL*/;
    const float *near_in, *far_in /*[R] placeholders or NULL (0 / 1)*/;
    int R, G, chunk /*rays per nan-mean chunk of the cylinder bounds (the reference: 4096)*/;
} DanboRays;

typedef struct DanboFrameOut {   /* the dict of render_rays: final maps, then the coarse pass' */
    float *rgb_map /*[R,3]*/, *disp_map, *acc_map /*[R]*/, *alpha, *weights /*[R,S+Sf]: alpha, T_i*/;
    float *rgb0, *disp0, *acc0, *alpha0 /*[R,S]*/;
} DanboFrameOut;

size_t danbo_render_frame_workspace(int R, int G, int S, int Sf, int chunk, int graph_width);
/* 3 <= S <= 256, Sf <= 64 (S <= 64: the fused composite + resampling launch).  Enqueues ~25 kernels on `stream`; workspace: danbo_render_frame_workspace bytes of device memory. */
int danbo_render_frame(const DanboModel* model, const DanboRays* rays, int S, int Sf, const DanboFrameOut* out, void* workspace,
                       size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DANBO_HIP_H */
